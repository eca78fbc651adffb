# coding: utf-8
"""torch.ops.dv3hip.* (deepvoice3_pytorch_amd/torch_ops.py: the operator surface of SURVEY.md 8b, registered with the
dispatcher) against the autograd.Functions the model classes call -- the face tests/test_gpu_kernels.py and
tests/test_gpu_model.py pin to the oracle.  Same kernels, same arguments: bit for bit, forward and every gradient.
Reference semantics: modules.py:112-229, conv.py:7-65, modules.py:103-109, deepvoice3.py:108-176, train.py:537-601,755-759."""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _dev():
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(_dev())


def _leaf(t):
    return t.clone().requires_grad_()


def _same(a, b):
    return a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("mode,causal,dil,p,spk", [(0, False, 1, 0.0, False), (0, True, 3, 0.05, False),
                                                   (1, False, 9, 0.05, False), (0, False, 1, 0.05, True)])
def test_conv1d_glu_operator_equals_the_module_path(mode, causal, dil, p, spk):
    from deepvoice3_pytorch_amd import ops, torch_ops
    B, C, T = 3, 64, 150
    x, v = _rand(B, C, T, seed=1), _rand(2 * C, C, 3, seed=2, scale=0.05)
    g, bias = _rand(2 * C, 1, 1, seed=3).abs() + 0.5, _rand(2 * C, seed=4, scale=0.1)
    sb = torch.tanh(_rand(B, C, seed=5)) if spk else None
    dy = _rand(B, C, T, seed=6)
    seed, site = 1234, 7
    # the module path (ops.conv_layer), masks drawn from the same (seed, site)
    a = [_leaf(t) for t in (x, v, g, bias)] + ([_leaf(sb)] if spk else [None])
    cfg = ops.LayerCfg(k=3, dil=dil, causal=causal, mode=torch_ops._MODES[mode], residual=(mode == 0), p=p, training=p > 0)
    with torch_ops._philox(seed, site):
        y0 = ops.conv_layer(a[0], a[1], a[2], a[3], cfg, spk=a[4])
    y0.backward(dy)
    b = [_leaf(t) for t in (x, v, g, bias)] + ([_leaf(sb)] if spk else [None])
    y1, pre, bits = torch.ops.dv3hip.conv1d_glu_fwd(b[0], b[1], b[2], b[3], b[4], dil, causal, mode, mode == 0, p, seed, site)
    assert pre.shape == (B, 2 * C, T) and (bits.numel() > 0) == (p > 0)
    assert _same(y1, y0)
    y1.backward(dy)
    for u, w in zip(a, b):
        if u is not None:
            assert _same(w.grad, u.grad)
    # the backward operator called directly
    gx, gv, gg, gb, gs = torch.ops.dv3hip.conv1d_glu_bwd(dy, x, v, g, sb, pre.detach(), bits, dil, causal, mode, mode == 0, p, True)
    assert _same(gx, a[0].grad) and _same(gv, a[1].grad) and _same(gg, a[2].grad) and _same(gb, a[3].grad)
    assert (gs.numel() == 0) == (not spk)


@pytest.mark.parametrize("k,pad,dil,act", [(1, 0, 1, 1), (1, 0, 1, 2), (3, 1, 1, 0), (5, 4, 2, 1)])
def test_conv1d_act_operator(k, pad, dil, act):
    from deepvoice3_pytorch_amd import ops, torch_ops
    B, Ci, Co, T = 2, 48, 80, 90
    x, v = _rand(B, Ci, T, seed=1), _rand(Co, Ci, k, seed=2, scale=0.1)
    g, bias = _rand(Co, 1, 1, seed=3).abs() + 0.5, _rand(Co, seed=4, scale=0.1)
    a = [_leaf(t) for t in (x, v, g, bias)]
    cfg = ops.LayerCfg(k=k, dil=dil, mode=torch_ops._ACTS[act])
    cfg.pad_left, cfg.t_out = pad, T + 2 * pad - dil * (k - 1)
    y0 = ops.conv_layer(a[0], a[1], a[2], a[3], cfg)
    dy = _rand(*y0.shape, seed=5)
    y0.backward(dy)
    b = [_leaf(t) for t in (x, v, g, bias)]
    y1 = torch.ops.dv3hip.conv1d_act_fwd(b[0], b[1], b[2], b[3], pad, dil, act)
    assert _same(y1, y0)
    y1.backward(dy)
    for u, w in zip(a, b):
        assert _same(w.grad, u.grad)


def test_convtranspose_operator():
    from deepvoice3_pytorch_amd import ops
    B, C, T = 2, 64, 50
    x, v = _rand(B, C, T, seed=1), _rand(C, C, 2, seed=2, scale=0.1)
    g, bias = _rand(C, 1, 1, seed=3).abs() + 0.5, _rand(C, seed=4, scale=0.1)
    a = [_leaf(t) for t in (x, v, g, bias)]
    y0 = ops.conv_layer(a[0], a[1], a[2], a[3], ops.LayerCfg(k=2, dil=1, mode=ops.EPI_LINEAR, transposed=True))
    dy = _rand(*y0.shape, seed=5)
    y0.backward(dy)
    b = [_leaf(t) for t in (x, v, g, bias)]
    y1 = torch.ops.dv3hip.convtranspose1d_k2s2_fwd(b[0], b[1], b[2], b[3])
    assert y1.shape == (B, C, 2 * T) and _same(y1, y0)
    y1.backward(dy)
    for u, w in zip(a, b):
        assert _same(w.grad, u.grad)


@pytest.mark.parametrize("p,masked", [(0.0, False), (0.1, True)])
def test_attention_operator(p, masked):
    from deepvoice3_pytorch_amd import ops, torch_ops
    B, E, Tq, Tk = 3, 64, 40, 29
    q, k, v = _rand(B, E, Tq, seed=1), _rand(B, E, Tk, seed=2), _rand(B, Tk, E, seed=3)
    kl = torch.tensor([29, 17, 9], dtype=torch.int32, device=_dev()) if masked else None
    dctx, dP = _rand(B, E, Tq, seed=4), _rand(B, Tq, Tk, seed=5, scale=0.01)
    a = [_leaf(t) for t in (q, k, v)]
    with torch_ops._philox(99, 3):
        c0, P0 = ops.attention_core(a[0], a[1], a[2].transpose(1, 2), kl, None, p, p > 0)
    torch.autograd.backward([c0, P0], [dctx, dP])
    b = [_leaf(t) for t in (q, k, v)]
    c1, P1, Pd, bits = torch.ops.dv3hip.attention_fwd(b[0], b[1], b[2], kl, p, 99, 3)
    assert _same(c1, c0) and _same(P1, P0)
    torch.autograd.backward([c1, P1], [dctx, dP])
    for u, w in zip(a, b):
        assert _same(w.grad, u.grad)
    if masked:
        assert float(P1[1, :, 17:].abs().max()) == 0.0
    assert abs(float(P1.sum()) - B * Tq) < 1e-3


def test_loss_operators_and_position_encoding():
    from deepvoice3_pytorch_amd import ops, modules
    dev = _dev()
    B, T, D = 3, 50, 20
    yh = (torch.rand(B, T, D, generator=torch.Generator().manual_seed(1)) * 0.98 + 0.01).to(dev)
    y = torch.rand(B, T, D, generator=torch.Generator().manual_seed(2)).to(dev)
    lens = torch.tensor([50, 31, 12], dtype=torch.int32, device=dev)
    a, b = _leaf(yh), _leaf(yh)
    o0 = ops.spec_loss(a, y, lens, 1, 0.5, 0.1)
    o0[2].backward()
    o1, gr = torch.ops.dv3hip.spec_loss_fwd(b, y, lens, 1, 0.5, 0.1)
    o1[2].backward()
    assert _same(o1, o0) and _same(b.grad, a.grad) and _same(gr, a.grad)
    attn = torch.rand(2, B, T, 13, generator=torch.Generator().manual_seed(3)).to(dev)
    il = torch.tensor([13, 9, 4], dtype=torch.int32, device=dev)
    a, b = _leaf(attn), _leaf(attn)
    l0 = ops.guided_attention_loss(a, il, lens, 0.2)
    l0[0].backward()
    l1, _ = torch.ops.dv3hip.guided_attn_loss_fwd(b, il, lens, 0.2)
    l1[0].backward()
    assert _same(l1, l0) and _same(b.grad, a.grad)
    p = (torch.rand(B, T, 1, generator=torch.Generator().manual_seed(4)) * 0.9 + 0.05).to(dev)
    t = (torch.rand(B, T, 1, generator=torch.Generator().manual_seed(5)) > 0.5).float().to(dev)
    a, b = _leaf(p), _leaf(p)
    l0 = ops.bce_loss(a, t)
    l0[0].backward()
    l1, _ = torch.ops.dv3hip.bce_loss_fwd(b, t)
    l1[0].backward()
    assert _same(l1, l0) and _same(b.grad, a.grad)
    # SinusoidalEncoding.forward (modules.py:45-64) at rate w
    enc = modules.SinusoidalEncoding(64, 32).to(dev)
    pos = torch.tensor([[1, 2, 3, 0, 0], [5, 6, 7, 8, 9]], dtype=torch.int64, device=dev)
    want = enc.forward_bct(pos, 1.29)
    got = torch.ops.dv3hip.sincos_pos_embed(pos, enc.weight, 1.29)
    assert _same(got, want)


def test_fused_clip_adam_operator_against_torch_adam():
    """clip_grad_norm_ + torch.optim.Adam on the same flat tensors (train.py:755-759)"""
    n = 10007
    p0, g0 = _rand(n, seed=1), _rand(n, seed=2, scale=0.01)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.5, 0.9), eps=1e-6)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for step in range(1, 4):
        gr = g0 * step
        ref.grad = gr.clone()
        total = torch.nn.utils.clip_grad_norm_([ref], 0.1)
        opt.step()
        norm = torch.ops.dv3hip.fused_clip_adam(p, gr, m, v, 1e-3, step, 0.5, 0.9, 1e-6, 0.0, 0.1)
        assert abs(float(norm) - float(total)) < 1e-5 * float(total)
        assert float((p - ref.detach()).abs().max()) < 2e-6


def test_audio_operators():
    from deepvoice3_pytorch_amd import audio
    mag = torch.rand(2, 40, 513, generator=torch.Generator().manual_seed(0)).to(_dev())
    w0 = audio.griffin_lim(mag, 256, 5)
    w1 = torch.ops.dv3hip.griffin_lim(mag, 256, 5)
    assert _same(w1, w0) and w1.dim() == 2
