# coding: utf-8
"""`torch.ops.dv3hip.*`: the operator surface of SURVEY.md 8b, registered with the PyTorch dispatcher.

north_star words the boundary as "hand-written CDNA4 HIP kernels exposed to Python via PyTorch-ROCm custom ops"; SURVEY 8b
lists the operators (`TORCH_LIBRARY(dv3hip, m)`: conv1d_glu_fwd / _bwd, conv1x1_act, convtranspose1d_k2s2, attention_fwd /
_bwd, sincos_pos_embed, the loss pairs, fused_clip_adam, griffin_lim, istft).  This module registers them from Python
(`torch.library.Library("dv3hip", "DEF")` is the Python face of the same registry TORCH_LIBRARY writes to): a schema per
operator, an implementation for the CUDA (= HIP on ROCm) dispatch key over the C ABI of include/dv3hip.h, a shape-only
implementation for meta tensors, and an autograd formula whose backward is itself a registered operator.  Operators are
FUNCTIONAL: tensors and scalars in, new tensors out; what a backward needs (the saved pre-gate pair, the dropout bits, the
probabilities) is an explicit output of the forward and an explicit input of the backward -- nothing hides in Python
objects, so a trace of a step through these operators shows every tensor that crosses the boundary.

The model classes (deepvoice3.py / nyanko.py) do NOT route through the dispatcher: a training step is ~330 launches and
the segment replay (train_step.GraphedTrainer) already removes the per-launch host cost; the dispatcher would add per-call
work without removing a launch.  Both faces call the same code: an operator here builds the arguments of the
autograd.Function the modules use (ops.ConvLayerFn, ops.AttnCoreFn, ...) and runs its forward / backward body, so the
numbers are the ones tests/test_gpu_*.py pin to the oracle; tests/test_gpu_torch_ops.py checks the two faces against each
other bit for bit.  There is no CPU implementation: a CPU tensor raises NotImplementedError from the dispatcher (the
product path has no fallback), meta tensors give shapes.

Reference semantics: modules.py:112-229 (Conv1dGLU / HighwayConv1d), conv.py:7-65, modules.py:103-109 (ConvTranspose1d),
deepvoice3.py:108-176 (AttentionLayer), modules.py:10-64 (SinusoidalEncoding), train.py:537-601,704-740 (losses),
train.py:755-759 (clip + Adam), audio.py:37-43 (inverse spectrogram)."""
import math

import torch

from . import ops

_NS = "dv3hip"
_lib = torch.library.Library(_NS, "DEF")

GLU, HIGHWAY = 0, 1                  # `mode` of conv1d_glu_*
ACT_LINEAR, ACT_RELU, ACT_SIGMOID = 0, 1, 2      # `act` of conv1d_act_*
_MODES = {GLU: ops.EPI_GLU, HIGHWAY: ops.EPI_HIGHWAY}
_ACTS = {ACT_LINEAR: ops.EPI_LINEAR, ACT_RELU: ops.EPI_RELU, ACT_SIGMOID: ops.EPI_SIGMOID}


class _Ctx(object):
    """what ops.*Fn.forward / .backward see as their autograd context when an operator runs their bodies"""

    def __init__(self, needs):
        self.needs_input_grad = tuple(needs)
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class _philox(object):
    """the dropout mask of an operator call is a function of its (philox_seed, philox_site) arguments: the process-wide
    ops.dropout_state (what the modules draw from) is set for the duration of the call and put back"""

    def __init__(self, seed, site):
        self.seed, self.site = int(seed), int(site)

    def __enter__(self):
        st = ops.dropout_state
        self.prev = (st.seed, st.site, st.dev_offset)
        st.seed, st.site, st.dev_offset = self.seed & 0xFFFFFFFFFFFFFFFF, self.site - 1, None
        return self

    def __exit__(self, *exc):
        st = ops.dropout_state
        st.seed, st.site, st.dev_offset = self.prev
        return False


def _empty(like, dtype=None):
    return torch.empty(0, dtype=dtype or like.dtype, device=like.device)


def _or_empty(t, like):
    return t if t is not None else _empty(like)


def _split_only(J, dil, T, Tout):
    return Tout == T and (J - 1) * dil <= 64 and J <= 16


# ------------------------------------------------------------------------------------------------------------------
# Conv1dGLU / HighwayConv1d (modules.py:112-229)
# ------------------------------------------------------------------------------------------------------------------
_lib.define("conv1d_glu_fwd(Tensor x, Tensor weight_v, Tensor? weight_g, Tensor? bias, Tensor? spk_bias, int dilation, "
            "bool causal, int mode, bool residual, float p_drop, int philox_seed, int philox_site) "
            "-> (Tensor y, Tensor pre_gate, Tensor mask_bits)")
_lib.define("conv1d_glu_bwd(Tensor grad_y, Tensor x, Tensor weight_v, Tensor? weight_g, Tensor? spk_bias, Tensor pre_gate, "
            "Tensor mask_bits, int dilation, bool causal, int mode, bool residual, float p_drop, bool has_bias) "
            "-> (Tensor grad_x, Tensor grad_v, Tensor grad_g, Tensor grad_bias, Tensor grad_spk)")


def _glu_cfg(v, dilation, causal, mode, residual, p_drop):
    if mode not in _MODES:
        raise ValueError("conv1d_glu: mode 0 (GLU) or 1 (highway)")
    return ops.LayerCfg(k=v.shape[2], dil=dilation, causal=causal, mode=_MODES[mode], residual=residual, p=p_drop,
                        training=p_drop > 0)


def _conv1d_glu_fwd(x, weight_v, weight_g, bias, spk_bias, dilation, causal, mode, residual, p_drop, philox_seed,
                    philox_site):
    cfg = _glu_cfg(weight_v, dilation, causal, mode, residual, p_drop)
    ctx = _Ctx((True,) * 9)
    with _philox(philox_seed, philox_site):
        y = ops.ConvLayerFn.forward(ctx, x, weight_v, weight_g, bias, spk_bias, None, None, cfg, None)
    pre = ctx.saved_tensors[3]
    bits = ctx.bits if ctx.bits is not None else _empty(x, torch.int32)
    if hasattr(y, "_dv3_tok"):
        del y._dv3_tok
    return y, pre, bits


def _conv1d_glu_bwd(grad_y, x, weight_v, weight_g, spk_bias, pre_gate, mask_bits, dilation, causal, mode, residual,
                    p_drop, has_bias):
    cfg = _glu_cfg(weight_v, dilation, causal, mode, residual, p_drop)
    B, Cin, T = x.shape
    O, _, J = weight_v.shape
    Cg = O // 2
    ctx = _Ctx((True, True, weight_g is not None, has_bias, spk_bias is not None, False, False, False, False))
    ctx.cfg = cfg
    ctx.pk = ops.pack_weights(weight_v, weight_g, glu_cg=Cg, need_bwd=True, split_only=_split_only(J, dilation, T, T))
    ctx.dims = (B, Cin, T, T, O, Cg, J, ops._pad_left(J, dilation, causal))
    has_mask = mask_bits.numel() > 0
    ctx.bits, ctx.bits_rs = (mask_bits, (T + 31) // 32) if has_mask else (None, 0)
    ctx.dscale = 1.0 / (1.0 - p_drop) if has_mask else 1.0
    ctx.spk_dim = spk_bias.dim() if spk_bias is not None else 0
    ctx.has_r = ctx.has_r2 = False
    ctx.has_bias, ctx.inplace, ctx.leaves, ctx.tok, ctx.prod, ctx.pair_ok = has_bias, False, None, None, None, False
    ctx.saved_tensors = (x.contiguous(), weight_v, weight_g, pre_gate)
    dx, dv, dg, dbias, dspk = ops.ConvLayerFn.backward(ctx, grad_y)[:5]
    if dspk is not None:
        dspk = dspk.contiguous()
    return dx, dv, _or_empty(dg, x), _or_empty(dbias if has_bias else None, x), _or_empty(dspk, x)


def _conv1d_glu_fwd_meta(x, weight_v, weight_g, bias, spk_bias, dilation, causal, mode, residual, p_drop, philox_seed,
                         philox_site):
    B, C, T = x.shape
    O = weight_v.shape[0]
    bits = x.new_empty((B * C * ((T + 31) // 32),) if p_drop > 0 else (0,), dtype=torch.int32)
    return x.new_empty((B, O // 2, T)), x.new_empty((B, O, T)), bits


def _conv1d_glu_bwd_meta(grad_y, x, weight_v, weight_g, spk_bias, pre_gate, mask_bits, dilation, causal, mode, residual,
                         p_drop, has_bias):
    e = x.new_empty((0,))
    return (torch.empty_like(x), torch.empty_like(weight_v), torch.empty_like(weight_g) if weight_g is not None else e,
            x.new_empty((weight_v.shape[0],)) if has_bias else e, torch.empty_like(spk_bias) if spk_bias is not None else e)


_lib.impl("conv1d_glu_fwd", _conv1d_glu_fwd, "CUDA")
_lib.impl("conv1d_glu_bwd", _conv1d_glu_bwd, "CUDA")
_lib.impl("conv1d_glu_fwd", _conv1d_glu_fwd_meta, "Meta")
_lib.impl("conv1d_glu_bwd", _conv1d_glu_bwd_meta, "Meta")


def _glu_setup(ctx, inputs, output):
    x, v, g, bias, spk, dilation, causal, mode, residual, p_drop, _, _ = inputs
    _, pre, bits = output
    ctx.save_for_backward(x, v, g, spk, pre, bits)
    ctx.args = (dilation, causal, mode, residual, p_drop, bias is not None)


def _glu_backward(ctx, gy, g_pre, g_bits):
    x, v, g, spk, pre, bits = ctx.saved_tensors
    dilation, causal, mode, residual, p_drop, has_bias = ctx.args
    dx, dv, dg, db, dspk = torch.ops.dv3hip.conv1d_glu_bwd(gy.contiguous(), x, v, g, spk, pre, bits, dilation, causal,
                                                           mode, residual, p_drop, has_bias)
    return (dx, dv, dg if g is not None else None, db if has_bias else None, dspk if spk is not None else None,
            None, None, None, None, None, None, None)


torch.library.register_autograd("dv3hip::conv1d_glu_fwd", _glu_backward, setup_context=_glu_setup)


# ------------------------------------------------------------------------------------------------------------------
# plain Conv1d (+ ReLU / sigmoid): conv.py:7-65, the 1 x 1 layers of SURVEY 8a row a4 are kernel size 1
# ------------------------------------------------------------------------------------------------------------------
_lib.define("conv1d_act_fwd(Tensor x, Tensor weight_v, Tensor? weight_g, Tensor? bias, int padding, int dilation, int act) "
            "-> Tensor")
_lib.define("conv1d_act_bwd(Tensor grad_y, Tensor x, Tensor weight_v, Tensor? weight_g, Tensor y, int padding, int dilation, "
            "int act, bool has_bias) -> (Tensor grad_x, Tensor grad_v, Tensor grad_g, Tensor grad_bias)")


def _act_cfg(v, T, padding, dilation, act):
    if act not in _ACTS:
        raise ValueError("conv1d_act: act 0 (linear), 1 (ReLU) or 2 (sigmoid)")
    k = v.shape[2] if v.dim() == 3 else 1
    cfg = ops.LayerCfg(k=k, dil=dilation, mode=_ACTS[act])
    cfg.pad_left, cfg.t_out = padding, T + 2 * padding - dilation * (k - 1)
    return cfg


def _conv1d_act_fwd(x, weight_v, weight_g, bias, padding, dilation, act):
    cfg = _act_cfg(weight_v, x.shape[2], padding, dilation, act)
    return ops.ConvLayerFn.forward(_Ctx((False,) * 9), x, weight_v, weight_g, bias, None, None, None, cfg, None)


def _conv1d_act_bwd(grad_y, x, weight_v, weight_g, y, padding, dilation, act, has_bias):
    B, Cin, T = x.shape
    cfg = _act_cfg(weight_v, T, padding, dilation, act)
    O = weight_v.shape[0]
    J = weight_v.shape[2] if weight_v.dim() == 3 else 1
    ctx = _Ctx((True, True, weight_g is not None, has_bias, False, False, False, False, False))
    ctx.cfg = cfg
    ctx.pk = ops.pack_weights(weight_v, weight_g, need_bwd=True, split_only=_split_only(J, dilation, T, cfg.t_out))
    ctx.dims = (B, Cin, T, cfg.t_out, O, 0, J, padding)
    ctx.bits, ctx.bits_rs, ctx.dscale, ctx.spk_dim = None, 0, 1.0, 0
    ctx.has_r = ctx.has_r2 = False
    ctx.has_bias, ctx.inplace, ctx.leaves, ctx.tok, ctx.prod, ctx.pair_ok = has_bias, False, None, None, None, False
    ctx.saved_tensors = (x.contiguous(), weight_v, weight_g, y)
    dx, dv, dg, dbias = ops.ConvLayerFn.backward(ctx, grad_y)[:4]
    return dx, dv, _or_empty(dg, x), _or_empty(dbias if has_bias else None, x)


def _conv1d_act_fwd_meta(x, weight_v, weight_g, bias, padding, dilation, act):
    k = weight_v.shape[2] if weight_v.dim() == 3 else 1
    return x.new_empty((x.shape[0], weight_v.shape[0], x.shape[2] + 2 * padding - dilation * (k - 1)))


_lib.impl("conv1d_act_fwd", _conv1d_act_fwd, "CUDA")
_lib.impl("conv1d_act_bwd", _conv1d_act_bwd, "CUDA")
_lib.impl("conv1d_act_fwd", _conv1d_act_fwd_meta, "Meta")


def _act_setup(ctx, inputs, output):
    x, v, g, bias, padding, dilation, act = inputs
    ctx.save_for_backward(x, v, g, output)
    ctx.args = (padding, dilation, act, bias is not None)


def _act_backward(ctx, gy):
    x, v, g, y = ctx.saved_tensors
    padding, dilation, act, has_bias = ctx.args
    dx, dv, dg, db = torch.ops.dv3hip.conv1d_act_bwd(gy.contiguous(), x, v, g, y, padding, dilation, act, has_bias)
    return dx, dv, dg if g is not None else None, db if has_bias else None, None, None, None


torch.library.register_autograd("dv3hip::conv1d_act_fwd", _act_backward, setup_context=_act_setup)


# ------------------------------------------------------------------------------------------------------------------
# ConvTranspose1d(kernel_size=2, stride=2) (modules.py:103-109; deepvoice3.py:519-520,527-528): exact x2 upsampling
# ------------------------------------------------------------------------------------------------------------------
_lib.define("convtranspose1d_k2s2_fwd(Tensor x, Tensor weight_v, Tensor? weight_g, Tensor? bias) -> Tensor")
_lib.define("convtranspose1d_k2s2_bwd(Tensor grad_y, Tensor x, Tensor weight_v, Tensor? weight_g, bool has_bias) "
            "-> (Tensor grad_x, Tensor grad_v, Tensor grad_g, Tensor grad_bias)")


def _convT_fwd(x, weight_v, weight_g, bias):
    cfg = ops.LayerCfg(k=2, dil=1, mode=ops.EPI_LINEAR, transposed=True)
    return ops.ConvLayerFn.forward(_Ctx((False,) * 9), x, weight_v, weight_g, bias, None, None, None, cfg, None)


def _convT_bwd(grad_y, x, weight_v, weight_g, has_bias):
    B, Cin, T = x.shape
    I, O, J = weight_v.shape
    cfg = ops.LayerCfg(k=2, dil=1, mode=ops.EPI_LINEAR, transposed=True)
    ctx = _Ctx((True, True, weight_g is not None, has_bias, False, False, False, False, False))
    ctx.cfg = cfg
    ctx.pk = ops.pack_weights(weight_v, weight_g, transposed=True, need_bwd=True)
    ctx.dims = (B, Cin, T, T, J * O, 0, J, 0)
    ctx.bits, ctx.bits_rs, ctx.dscale, ctx.spk_dim = None, 0, 1.0, 0
    ctx.has_r = ctx.has_r2 = False
    ctx.has_bias, ctx.inplace, ctx.leaves, ctx.tok, ctx.prod, ctx.pair_ok = has_bias, False, None, None, None, False
    ctx.saved_tensors = (x.contiguous(), weight_v, weight_g, None)
    dx, dv, dg, dbias = ops.ConvLayerFn.backward(ctx, grad_y)[:4]
    return dx, dv, _or_empty(dg, x), _or_empty(dbias if has_bias else None, x)


_lib.impl("convtranspose1d_k2s2_fwd", _convT_fwd, "CUDA")
_lib.impl("convtranspose1d_k2s2_bwd", _convT_bwd, "CUDA")
_lib.impl("convtranspose1d_k2s2_fwd", lambda x, v, g, b: x.new_empty((x.shape[0], v.shape[1], 2 * x.shape[2])), "Meta")


def _convT_setup(ctx, inputs, output):
    x, v, g, bias = inputs
    ctx.save_for_backward(x, v, g)
    ctx.has_bias = bias is not None


def _convT_backward(ctx, gy):
    x, v, g = ctx.saved_tensors
    dx, dv, dg, db = torch.ops.dv3hip.convtranspose1d_k2s2_bwd(gy.contiguous(), x, v, g, ctx.has_bias)
    return dx, dv, dg if g is not None else None, db if ctx.has_bias else None


torch.library.register_autograd("dv3hip::convtranspose1d_k2s2_fwd", _convT_backward, setup_context=_convT_setup)


# ------------------------------------------------------------------------------------------------------------------
# attention core (deepvoice3.py:143-171): q (B, E, Tq), keysT (B, E, Tk) as the reference pre-transposes them
# (deepvoice3.py:318), values (B, Tk, E)
# ------------------------------------------------------------------------------------------------------------------
_lib.define("attention_fwd(Tensor q, Tensor keysT, Tensor values, Tensor? key_pad_lengths, float p_drop, int philox_seed, "
            "int philox_site) -> (Tensor ctx, Tensor P, Tensor Pd, Tensor mask_bits)")
_lib.define("attention_bwd(Tensor grad_ctx, Tensor? grad_P, Tensor q, Tensor keysT, Tensor values, Tensor P, Tensor Pd, "
            "Tensor mask_bits, float p_drop) -> (Tensor grad_q, Tensor grad_keysT, Tensor grad_values)")


def _attention_fwd(q, keysT, values, key_pad_lengths, p_drop, philox_seed, philox_site):
    ctx = _Ctx((True, True, True, False, False, False))
    v_bct = values.transpose(1, 2).contiguous()
    with _philox(philox_seed, philox_site):
        c, P = ops.AttnCoreFn.forward(ctx, q, keysT, v_bct, key_pad_lengths, None, (p_drop, p_drop > 0, 1, 3, None))
    pd = ctx.saved_tensors[4]
    return c, P, pd, ctx.bits if ctx.bits is not None else _empty(q, torch.int32)


def _attention_bwd(grad_ctx, grad_P, q, keysT, values, P, Pd, mask_bits, p_drop):
    Tk = keysT.shape[2]
    ctx = _Ctx((True, True, True, False, False, False))
    ctx.saved_tensors = (q.contiguous(), keysT.contiguous(), values.transpose(1, 2).contiguous(), P, Pd)
    has_mask = mask_bits.numel() > 0
    ctx.bits, ctx.bits_rs = (mask_bits, (Tk + 31) // 32) if has_mask else (None, 0)
    ctx.dscale = 1.0 / (1.0 - p_drop) if has_mask else 1.0
    ctx.pd_scale, ctx.scale_dev = Tk * math.sqrt(1.0 / Tk), None
    dq, dk, dv = ops.AttnCoreFn.backward(ctx, grad_ctx, grad_P)[:3]
    return dq, dk, dv.transpose(1, 2).contiguous()


def _attention_fwd_meta(q, keysT, values, key_pad_lengths, p_drop, philox_seed, philox_site):
    B, E, Tq = q.shape
    Tk = keysT.shape[2]
    bits = q.new_empty((B * Tq * ((Tk + 31) // 32),) if p_drop > 0 else (0,), dtype=torch.int32)
    return q.new_empty((B, E, Tq)), q.new_empty((B, Tq, Tk)), q.new_empty((B, Tq, Tk)), bits


_lib.impl("attention_fwd", _attention_fwd, "CUDA")
_lib.impl("attention_bwd", _attention_bwd, "CUDA")
_lib.impl("attention_fwd", _attention_fwd_meta, "Meta")


def _attn_setup(ctx, inputs, output):
    q, k, v, _, p_drop, _, _ = inputs
    _, P, Pd, bits = output
    ctx.save_for_backward(q, k, v, P, Pd, bits)
    ctx.p_drop = p_drop


def _attn_backward(ctx, g_ctx, g_P, g_Pd, g_bits):
    q, k, v, P, Pd, bits = ctx.saved_tensors
    if g_ctx is None:
        g_ctx = torch.zeros_like(q)
    dq, dk, dv = torch.ops.dv3hip.attention_bwd(g_ctx.contiguous(), g_P, q, k, v, P, Pd, bits, ctx.p_drop)
    return dq, dk, dv, None, None, None, None


torch.library.register_autograd("dv3hip::attention_fwd", _attn_backward, setup_context=_attn_setup)


# ------------------------------------------------------------------------------------------------------------------
# SinusoidalEncoding.forward (modules.py:45-64): positions (B, T) int64, table (n_pos, C) raw angles -> (B, C, T)
# ------------------------------------------------------------------------------------------------------------------
_lib.define("sincos_pos_embed(Tensor positions, Tensor table, float w) -> Tensor")
_lib.impl("sincos_pos_embed", lambda positions, table, w: ops.sincos_pos_bct(positions, table, float(w)), "CUDA")
_lib.impl("sincos_pos_embed", lambda positions, table, w: table.new_empty((positions.shape[0], table.shape[1],
                                                                           positions.shape[1])), "Meta")


# ------------------------------------------------------------------------------------------------------------------
# losses: value and gradient in one pass (train.py:537-601,704-740)
# ------------------------------------------------------------------------------------------------------------------
_lib.define("spec_loss_fwd(Tensor y_hat, Tensor y, Tensor? lengths, int r, float w_masked, float w_bd) "
            "-> (Tensor out4, Tensor grad_y_hat)")
_lib.define("guided_attn_loss_fwd(Tensor attn, Tensor in_lens, Tensor out_lens, float g) -> (Tensor loss, Tensor grad_attn)")
_lib.define("bce_loss_fwd(Tensor p, Tensor target) -> (Tensor loss, Tensor grad_p)")
_lib.impl("spec_loss_fwd", lambda y_hat, y, lengths, r, w_masked, w_bd: ops.spec_loss_with_grad(y_hat, y, lengths, r, w_masked,
                                                                                                 w_bd), "CUDA")
_lib.impl("guided_attn_loss_fwd", lambda attn, il, ol, g: ops.guided_attention_loss_with_grad(attn, il, ol, g), "CUDA")
_lib.impl("bce_loss_fwd", lambda p, t: ops.bce_loss_with_grad(p, t), "CUDA")
_lib.impl("spec_loss_fwd", lambda y_hat, y, lengths, r, w_masked, w_bd: (y_hat.new_empty((4,)), torch.empty_like(y_hat)), "Meta")
_lib.impl("guided_attn_loss_fwd", lambda attn, il, ol, g: (attn.new_empty((1,)), torch.empty_like(attn)), "Meta")
_lib.impl("bce_loss_fwd", lambda p, t: (p.new_empty((1,)), torch.empty_like(p)), "Meta")


def _loss_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])


def _spec_backward(ctx, g_out4, g_grad):
    (gr,) = ctx.saved_tensors
    return gr * g_out4[2], None, None, None, None, None      # only the total (out4[2]) carries gradient (ops.SpecLossFn)


def _scalar_backward(n_inputs):
    def backward(ctx, g_loss, g_grad):
        (gr,) = ctx.saved_tensors
        return (gr * g_loss,) + (None,) * (n_inputs - 1)
    return backward


torch.library.register_autograd("dv3hip::spec_loss_fwd", _spec_backward, setup_context=_loss_setup)
torch.library.register_autograd("dv3hip::guided_attn_loss_fwd", _scalar_backward(4), setup_context=_loss_setup)
torch.library.register_autograd("dv3hip::bce_loss_fwd", _scalar_backward(2), setup_context=_loss_setup)


# ------------------------------------------------------------------------------------------------------------------
# clip_grad_norm_ + Adam over flat arenas (train.py:755-759), in place; -> the gradient's 2-norm before clipping
# ------------------------------------------------------------------------------------------------------------------
_lib.define("fused_clip_adam(Tensor(a!) params, Tensor grads, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, float lr, int step, "
            "float beta1, float beta2, float eps, float weight_decay, float max_norm) -> Tensor")


def _fused_clip_adam(params, grads, exp_avg, exp_avg_sq, lr, step, beta1, beta2, eps, weight_decay, max_norm):
    dev = params.device
    hyper = torch.tensor([lr, 1.0 - beta1 ** step, math.sqrt(1.0 - beta2 ** step)], dtype=torch.float32).to(dev)
    norm = torch.empty(2, dtype=torch.float32, device=dev)
    partial = torch.empty(1024, dtype=torch.float32, device=dev)
    ops.grad_sqnorm(grads, partial, norm)
    ops.clip_adam(params, grads, exp_avg, exp_avg_sq, norm if max_norm > 0 else None, max_norm, hyper, beta1, beta2, eps,
                  weight_decay)
    return norm[0:1].clone()


_lib.impl("fused_clip_adam", _fused_clip_adam, "CUDA")


# ------------------------------------------------------------------------------------------------------------------
# audio.inv_spectrogram's phase reconstruction (audio.py:37-43; csrc/audio.hip): magnitudes (B, T, n_fft/2+1)
# ------------------------------------------------------------------------------------------------------------------
_lib.define("griffin_lim(Tensor mag, int hop, int n_iter) -> Tensor")
_lib.define("istft(Tensor mag, Tensor phasor, int hop) -> Tensor")


def _griffin_lim(mag, hop, n_iter):
    from . import audio
    return audio.griffin_lim(mag, hop, n_iter)


def _istft(mag, phasor, hop):
    from . import audio
    return audio.istft(mag, phasor, hop)


_lib.impl("griffin_lim", _griffin_lim, "CUDA")
_lib.impl("istft", _istft, "CUDA")

OPERATORS = ("conv1d_glu_fwd", "conv1d_glu_bwd", "conv1d_act_fwd", "conv1d_act_bwd", "convtranspose1d_k2s2_fwd",
             "convtranspose1d_k2s2_bwd", "attention_fwd", "attention_bwd", "sincos_pos_embed", "spec_loss_fwd",
             "guided_attn_loss_fwd", "bce_loss_fwd", "fused_clip_adam", "griffin_lim", "istft")
