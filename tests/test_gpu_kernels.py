# coding: utf-8
"""-m gpu: kernel-level parity of the HIP path (through the C ABI) against the CPU oracle.

Tolerance: BASELINE.json's north_star asks for 1e-4 relative fp32 on the model outputs; single
kernels are held to 2e-5 (fp32 MFMA is an exact fma chain; only the summation order differs from
the CPU's blocked GEMMs).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dv3_oracle as O
from tests.util import rel_err

pytestmark = pytest.mark.gpu

# per-kernel tolerance (max |err| / max |ref|): the exact fp32 MFMA chain, and the split-bf16
# ("bf16x3") operand form whose dropped lo*lo terms cost ~3*2^-18 per product (include/dv3hip.h)
KTOL_BY_MODE = {"f32": 2e-5, "bf16x3": 5e-5, "f16x3": 5e-5}   # f16x3: fp16-split forward (~1e-6), bf16-split gradients
KTOL = 2e-5


@pytest.fixture(autouse=True, params=["f16x3", "bf16x3", "f32"])
def gemm_mode(request):
    """every test in this module runs under both GEMM arithmetic modes"""
    global KTOL
    from deepvoice3_pytorch_amd import ops
    prev = ops.set_gemm_precision(request.param)
    KTOL = KTOL_BY_MODE[request.param]
    yield request.param
    ops.set_gemm_precision(prev)
    KTOL = 2e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


KS_DEFAULT = 1      # dv3_debug_set(44, v): the k-split form of the 128 x 64 tile (conv_gemm_bf16x3.hip): 0 off / 1 by rule / 2 wherever eligible
DP_DEFAULT = 0      # dv3_debug_set(43, v): the deep-prefetch form of the 128 x 64 tile is off (conv_gemm_bf16x3.hip, experiment build)


def _ops():
    from deepvoice3_pytorch_amd import ops
    return ops


def test_library_loads_on_gpu_box(dev):
    from deepvoice3_pytorch_amd import _lib
    h = _lib.lib()
    name = torch.zeros(1)  # placeholder to keep torch imported first
    import ctypes
    buf = ctypes.create_string_buffer(256)
    ncu = ctypes.c_int(0)
    _lib.call("dv3_device_info", 0, buf, 256, ctypes.byref(ncu))
    assert buf.value.decode().startswith("gfx950"), buf.value
    assert ncu.value == 256


def test_ops_refuse_cpu_tensors():
    ops = _ops()
    with pytest.raises(RuntimeError):
        ops.conv_layer(torch.zeros(1, 4, 8), torch.zeros(8, 4, 3), None, None, ops.LayerCfg(k=3))


def _glu_sd(C, k, rng, n_spk=0):
    sd = {"l.conv.weight_v": torch.from_numpy(rng.randn(2 * C, C, k).astype(np.float32) * 0.2),
          "l.conv.weight_g": torch.from_numpy(rng.uniform(0.5, 1.5, (2 * C, 1, 1)).astype(np.float32)),
          "l.conv.bias": torch.from_numpy(rng.uniform(-0.2, 0.2, 2 * C).astype(np.float32))}
    return sd


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 11, 12, 13, 14, 15, 16])
@pytest.mark.parametrize("C,T,k,d,causal", [(64, 200, 3, 1, False), (96, 150, 3, 27, True),
                                            (20, 37, 5, 3, False), (128, 513, 3, 9, True)])
def test_conv_gemm_glu_forward(dev, tile, C, T, k, d, causal):
    """Conv1dGLU forward, eval mode, every tile configuration (asymmetric random data)."""
    ops = _ops()
    rng = np.random.RandomState(C + T + k + d)
    B = 3
    sd = _glu_sd(C, k, rng)
    x = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
    for residual in (True, False):
        want = O.conv1d_glu(sd, "l", x, k, d, causal, residual)
        pk = ops.pack_weights(sd["l.conv.weight_v"].to(dev), sd["l.conv.weight_g"].to(dev), glu_cg=C,
                              need_bwd=False)
        padL = (k - 1) * d if causal else (k - 1) // 2 * d
        xg = x.to(dev)
        y = ops.conv_gemm(xg, pk.fwd, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d,
                          padL=padL, mode=ops.EPI_GLU, Cg=C, bias=sd["l.conv.bias"].to(dev),
                          r=xg if residual else None, residual=int(residual), tile_hint=tile)
        assert rel_err(y.cpu(), want) < KTOL


@pytest.mark.parametrize("tile", [0, 21, 22, 23, 24, 25, 26, 27, 28, 29])
@pytest.mark.parametrize("B,C,T,k,d,causal", [(3, 64, 200, 3, 1, False), (3, 96, 150, 3, 27, True),
                                              (5, 24, 37, 5, 3, False), (2, 128, 513, 3, 9, True),
                                              (7, 40, 50, 2, 4, True),
                                              # degenerate sizes: one frame, sequences shorter than the
                                              # receptive field, a 1x1 conv
                                              (1, 8, 1, 3, 1, True), (1, 16, 3, 3, 2, False),
                                              (2, 8, 33, 1, 1, False)])
def test_conv_gemm_bf16x3_forward(dev, gemm_mode, tile, B, C, T, k, d, causal):
    """the split-bf16 tap-GEMM, every tile: column tiles span several batch items here (B*T is
    flattened), so the per-fragment sequence-edge zeroing is exercised for every tap"""
    if gemm_mode == "f32":
        pytest.skip("split-operand kernel test")
    ops = _ops()
    rng = np.random.RandomState(C + T + k + d)
    sd = _glu_sd(C, k, rng)
    x = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
    pk = ops.pack_weights(sd["l.conv.weight_v"].to(dev), sd["l.conv.weight_g"].to(dev), glu_cg=C, need_bwd=False)
    assert pk.fwd_s is not None
    padL = (k - 1) * d if causal else (k - 1) // 2 * d
    xg = x.to(dev)
    for residual in (True, False):
        want = O.conv1d_glu(sd, "l", x, k, d, causal, residual)
        try:
            y = ops.conv_gemm(xg, pk.fwd, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d,
                              padL=padL, mode=ops.EPI_GLU, Cg=C, bias=sd["l.conv.bias"].to(dev),
                              r=xg if residual else None, residual=int(residual), tile_hint=tile, a_split=pk.fwd_s)
        except RuntimeError as e:
            assert tile != 0 and "needs split-bf16" in str(e)   # forced tile not eligible (LDS): fine
            pytest.skip("tile %d not eligible for this shape" % tile)
        # the scaled-fp16 split is an fp32-class operand form (2^-22): held to 5e-6, the bf16 split to 5e-5
        assert rel_err(y.cpu(), want) < (5e-6 if gemm_mode == "f16x3" else KTOL)


@pytest.mark.parametrize("tile", [28, 29])
@pytest.mark.parametrize("B,C,T,k,d,causal", [(3, 96, 150, 3, 27, True), (2, 256, 600, 3, 1, False),
                                              (5, 24, 37, 5, 3, False), (4, 160, 300, 1, 1, False)])
@pytest.mark.parametrize("masked", [False, True])
def test_conv_gemm_bf16x3_pingpong_equals_inphase(dev, gemm_mode, tile, B, C, T, k, d, causal, masked):
    """the 8-wave tiles run a ping-pong main loop (SIMD partners half a step apart); it accumulates
    in the order of the in-phase loop, so the two must agree bit for bit -- with the oracle as the
    anchor of one of them.  Covers several chunks (C > 32), partial chunks, 1x1 and 5 taps, the
    dropout keep-bits path."""
    if gemm_mode == "f32":
        pytest.skip("split-operand kernel test")
    ops = _ops()
    from deepvoice3_pytorch_amd import _lib
    rng = np.random.RandomState(C + T + k + d)
    sd = _glu_sd(C, k, rng)
    x = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
    pk = ops.pack_weights(sd["l.conv.weight_v"].to(dev), sd["l.conv.weight_g"].to(dev), glu_cg=C, need_bwd=False)
    padL = (k - 1) * d if causal else (k - 1) // 2 * d
    xg = x.to(dev)
    kw = {}
    if masked:
        ops.dropout_state.manual_seed(11)
        bits, rs = ops.dropout_bits(B * C, T, 0.25, dev)
        kw = dict(xmask=bits, xmask_rs=rs, drop_scale=1 / 0.75)
    ys = []
    try:
        for mode in (0, 1):
            _lib.call("dv3_debug_set", 3, mode)
            ys.append(ops.conv_gemm(xg, pk.fwd, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d,
                                    padL=padL, mode=ops.EPI_GLU, Cg=C, bias=sd["l.conv.bias"].to(dev), r=xg,
                                    residual=1, tile_hint=tile, a_split=pk.fwd_s, **kw))
    except RuntimeError as e:
        assert "needs split-bf16" in str(e)
        pytest.skip("tile %d not eligible for this shape" % tile)
    finally:
        _lib.call("dv3_debug_set", 3, 1)
    assert torch.equal(ys[0], ys[1])
    if not masked:
        assert rel_err(ys[1].cpu(), O.conv1d_glu(sd, "l", x, k, d, causal, True)) < KTOL


def test_conv_gemm_highway_and_activations(dev):
    ops = _ops()
    rng = np.random.RandomState(5)
    B, C, T, k, d = 2, 48, 77, 3, 3
    sd = _glu_sd(C, k, rng)
    x = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
    want = O.highway_conv1d(sd, "l", x, k, d, True)
    xg = x.to(dev)
    pk = ops.pack_weights(sd["l.conv.weight_v"].to(dev), sd["l.conv.weight_g"].to(dev), glu_cg=C, need_bwd=False)
    y = ops.conv_gemm(xg, pk.fwd, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d,
                      padL=(k - 1) * d, mode=ops.EPI_HIGHWAY, Cg=C, bias=sd["l.conv.bias"].to(dev), r=xg)
    assert rel_err(y.cpu(), want) < KTOL
    # 1x1 conv with odd channel counts (80 -> 513, the converter's last layer shape class)
    w = torch.from_numpy(rng.randn(513, 80, 1).astype(np.float32) * 0.1)
    b = torch.from_numpy(rng.randn(513).astype(np.float32) * 0.1)
    x2 = torch.from_numpy(rng.randn(B, 80, 203).astype(np.float32))
    pk2 = ops.pack_weights(w.to(dev), None, need_bwd=False)
    for mode, fn in [(ops.EPI_LINEAR, lambda v: v), (ops.EPI_RELU, torch.relu), (ops.EPI_SIGMOID, torch.sigmoid),
                     (ops.EPI_SOFTSIGN, F.softsign)]:
        y2 = ops.conv_gemm(x2.to(dev), pk2.fwd, pk2.lda, 0, B=B, Cin=80, Tin=203, M=513, Tout=203, mode=mode,
                           bias=b.to(dev))
        assert rel_err(y2.cpu(), fn(F.conv1d(x2, w, b))) < KTOL


def _grads(outs, ins):
    return torch.autograd.grad(outs, ins, allow_unused=True)


@pytest.mark.parametrize("T", [61, 64])     # 64: rows 16-byte aligned -> the 4-wide gate backward
@pytest.mark.parametrize("kind", ["glu_res", "glu", "highway", "relu1x1", "linear", "sigmoid", "convT"])
def test_conv_layer_backward(dev, kind, T):
    """ConvLayerFn forward+backward (pack -> tap-GEMM -> gate bwd -> dgrad -> wgrad -> weight-norm
    bwd) against torch autograd of the oracle's formulation."""
    ops = _ops()
    rng = np.random.RandomState(sum(ord(c) for c in kind))
    B, C, k, d = 4, 40, 3, 3
    if kind in ("glu_res", "glu", "highway"):
        v = torch.from_numpy(rng.randn(2 * C, C, k).astype(np.float32) * 0.2)
        g = torch.from_numpy(rng.uniform(0.5, 1.5, (2 * C, 1, 1)).astype(np.float32))
        bias = torch.from_numpy(rng.uniform(-0.2, 0.2, 2 * C).astype(np.float32))
    elif kind == "convT":
        v = torch.from_numpy(rng.randn(C, 24, 2).astype(np.float32) * 0.2)
        g = torch.from_numpy(rng.uniform(0.5, 1.5, (C, 1, 1)).astype(np.float32))
        bias = torch.from_numpy(rng.uniform(-0.2, 0.2, 24).astype(np.float32))
    else:
        v = torch.from_numpy(rng.randn(33, C, 1).astype(np.float32) * 0.2)
        g = torch.from_numpy(rng.uniform(0.5, 1.5, (33, 1, 1)).astype(np.float32))
        bias = torch.from_numpy(rng.uniform(-0.2, 0.2, 33).astype(np.float32))
    x = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))

    def cpu_forward(x, v, g, bias):
        sd = {"l.conv.weight_v": v, "l.conv.weight_g": g, "l.conv.bias": bias,
              "c.weight_v": v, "c.weight_g": g, "c.bias": bias}
        if kind == "glu_res":
            return O.conv1d_glu(sd, "l", x, k, d, True, True)
        if kind == "glu":
            return O.conv1d_glu(sd, "l", x, k, d, False, False)
        if kind == "highway":
            return O.highway_conv1d(sd, "l", x, k, d, True)
        if kind == "convT":
            return O.conv_transpose1d_k2s2(sd, "c", x)
        y = O.conv1d(sd, "c", x)
        return {"relu1x1": torch.relu, "linear": lambda t: t, "sigmoid": torch.sigmoid}[kind](y)

    cfg = {"glu_res": ops.LayerCfg(k=k, dil=d, causal=True, mode=ops.EPI_GLU, residual=True),
           "glu": ops.LayerCfg(k=k, dil=d, causal=False, mode=ops.EPI_GLU, residual=False),
           "highway": ops.LayerCfg(k=k, dil=d, causal=True, mode=ops.EPI_HIGHWAY),
           "relu1x1": ops.LayerCfg(mode=ops.EPI_RELU), "linear": ops.LayerCfg(mode=ops.EPI_LINEAR),
           "sigmoid": ops.LayerCfg(mode=ops.EPI_SIGMOID),
           "convT": ops.LayerCfg(k=2, mode=ops.EPI_LINEAR, transposed=True)}[kind]

    cin = [t.clone().requires_grad_(True) for t in (x, v, g, bias)]
    yc = cpu_forward(*cin)
    wgt = torch.from_numpy(rng.randn(*yc.shape).astype(np.float32))
    gc = _grads((yc * wgt).sum(), cin)

    gin = [t.clone().to(dev).requires_grad_(True) for t in (x, v, g, bias)]
    yg = ops.conv_layer(gin[0], gin[1], gin[2], gin[3], cfg)
    assert rel_err(yg.detach().cpu(), yc.detach()) < KTOL
    gg = _grads((yg * wgt.to(dev)).sum(), gin)
    for name, a, b in zip(("dx", "dv", "dg", "dbias"), gg, gc):
        assert rel_err(a.cpu(), b) < 5e-5, name


def test_conv_layer_dropout_and_residuals(dev):
    """Training-mode dropout (Philox keep-bits replayed into the oracle) + fused r / r2 residuals."""
    ops = _ops()
    rng = np.random.RandomState(9)
    B, C, T, k, d, p = 3, 32, 100, 3, 1, 0.25
    v = torch.from_numpy(rng.randn(2 * C, C, k).astype(np.float32) * 0.2)
    g = torch.from_numpy(rng.uniform(0.5, 1.5, (2 * C, 1, 1)).astype(np.float32))
    bias = torch.from_numpy(rng.uniform(-0.2, 0.2, 2 * C).astype(np.float32))
    x = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
    ops.dropout_state.manual_seed(77)
    ops.dropout_state.record = {}
    gin = [t.clone().to(dev).requires_grad_(True) for t in (x, v, g, bias)]
    cfg = ops.LayerCfg(k=k, dil=d, causal=False, mode=ops.EPI_GLU, residual=True, p=p, training=True, site="s0")
    yg = ops.conv_layer(gin[0], gin[1], gin[2], gin[3], cfg)
    bits, rows, TT = ops.dropout_state.record["s0"]
    ops.dropout_state.record = None
    keep = torch.from_numpy(O.unpack_keep_bits(bits.cpu().numpy().view(np.uint32), rows, (TT + 31) // 32, TT))
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.02
    # bit-exact agreement of the generator with the numpy Philox
    want_bits = O.philox_keep_bits(bits.numel(), p, 77, 1)
    assert np.array_equal(bits.cpu().numpy().view(np.uint32), want_bits)

    def drop(site, t, pp, layout):
        return t * keep.view(B, C, T).float() / (1 - pp)
    cin = [t.clone().requires_grad_(True) for t in (x, v, g, bias)]
    sd = {"l.conv.weight_v": cin[1], "l.conv.weight_g": cin[2], "l.conv.bias": cin[3]}
    yc = O.conv1d_glu(sd, "l", cin[0], k, d, False, True, p, drop)
    assert rel_err(yg.detach().cpu(), yc.detach()) < KTOL
    wgt = torch.from_numpy(rng.randn(*yc.shape).astype(np.float32))
    gc = _grads((yc * wgt).sum(), cin)
    gg = _grads((yg * wgt.to(dev)).sum(), gin)
    for name, a, b in zip(("dx", "dv", "dg", "dbias"), gg, gc):
        assert rel_err(a.cpu(), b) < 5e-5, name
    # linear + two fused residuals: y = ((Wx+b + r)*s + r2)*s
    w = torch.from_numpy(rng.randn(C, C, 1).astype(np.float32) * 0.2)
    r = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
    r2 = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
    b1 = bias[:C].clone()
    cin = [t.clone().requires_grad_(True) for t in (x, w, b1, r, r2)]
    s = math.sqrt(0.5)
    yc = ((F.conv1d(cin[0], cin[1], cin[2]) + cin[3]) * s + cin[4]) * s
    gin = [t.clone().to(dev).requires_grad_(True) for t in (x, w, b1, r, r2)]
    yg = ops.conv_layer(gin[0], gin[1], None, gin[2], ops.LayerCfg(), r=gin[3], r2=gin[4])
    assert rel_err(yg.detach().cpu(), yc.detach()) < KTOL
    gc = _grads((yc * wgt).sum(), cin)
    gg = _grads((yg * wgt.to(dev)).sum(), gin)
    for name, a, b in zip(("dx", "dw", "db", "dr", "dr2"), gg, gc):
        assert rel_err(a.cpu(), b) < 5e-5, name


def test_speaker_bias_paths(dev):
    ops = _ops()
    rng = np.random.RandomState(21)
    B, C, T, k = 3, 24, 50, 3
    v = torch.from_numpy(rng.randn(2 * C, C, k).astype(np.float32) * 0.2)
    bias = torch.from_numpy(rng.uniform(-0.2, 0.2, 2 * C).astype(np.float32))
    x = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
    for shape in [(B, C), (B, C, T)]:
        spk = torch.from_numpy(rng.randn(*shape).astype(np.float32))
        cin = [t.clone().requires_grad_(True) for t in (x, v, bias, spk)]
        y = F.conv1d(cin[0], cin[1], cin[2], padding=1)
        a, b = y.split(C, dim=1)
        a = a + (cin[3].unsqueeze(-1) if len(shape) == 2 else cin[3])
        yc = a * torch.sigmoid(b)
        gin = [t.clone().to(dev).requires_grad_(True) for t in (x, v, bias, spk)]
        yg = ops.conv_layer(gin[0], gin[1], None, gin[2], ops.LayerCfg(k=k, mode=ops.EPI_GLU), spk=gin[3])
        assert rel_err(yg.detach().cpu(), yc.detach()) < KTOL
        wgt = torch.from_numpy(rng.randn(*yc.shape).astype(np.float32))
        gc = _grads((yc * wgt).sum(), cin)
        gg = _grads((yg * wgt.to(dev)).sum(), gin)
        for name, u, w_ in zip(("dx", "dv", "db", "dspk"), gg, gc):
            assert rel_err(u.cpu(), w_) < 5e-5, name


@pytest.mark.parametrize("Tq,Tk,E", [(50, 37, 32), (203, 150, 64), (1, 29, 48)])
def test_attention_core(dev, Tq, Tk, E):
    """q^T k -> mask -> softmax -> dropout(off) -> context * sqrt(Tk), forward and backward."""
    ops = _ops()
    rng = np.random.RandomState(Tq + Tk)
    B = 3
    q, k, v = [torch.from_numpy(rng.randn(B, E, t).astype(np.float32) * 0.5) for t in (Tq, Tk, Tk)]
    lens = torch.tensor([Tk, max(1, Tk - 5), max(1, Tk // 2)], dtype=torch.int32)
    cin = [t.clone().requires_grad_(True) for t in (q, k, v)]
    S = torch.bmm(cin[0].transpose(1, 2), cin[1])
    mask = torch.arange(Tk)[None, :] >= lens[:, None]
    S = S.masked_fill(mask[:, None, :], -float("inf"))
    P = F.softmax(S, dim=-1)
    ctx = torch.bmm(P, cin[2].transpose(1, 2)) * (Tk * math.sqrt(1.0 / Tk))   # (B,Tq,E)
    ctx = ctx.transpose(1, 2)
    gin = [t.clone().to(dev).requires_grad_(True) for t in (q, k, v)]
    cg, Pg = ops.attention_core(gin[0], gin[1], gin[2], lens.to(dev))
    assert rel_err(Pg.detach().cpu(), P.detach()) < KTOL
    assert rel_err(cg.detach().cpu(), ctx.detach()) < KTOL
    w1 = torch.from_numpy(rng.randn(*ctx.shape).astype(np.float32))
    w2 = torch.from_numpy(rng.randn(*P.shape).astype(np.float32))
    gc = _grads((ctx * w1).sum() + (P * w2).sum(), cin)
    gg = _grads((cg * w1.to(dev)).sum() + (Pg * w2.to(dev)).sum(), gin)
    for name, a, b in zip(("dq", "dk", "dv"), gg, gc):
        assert rel_err(a.cpu(), b) < 1e-4, name


def test_attention_window_and_argmax(dev):
    ops = _ops()
    rng = np.random.RandomState(3)
    B, E, Tk = 2, 16, 40
    q = torch.from_numpy(rng.randn(B, E, 1).astype(np.float32))
    k = torch.from_numpy(rng.randn(B, E, Tk).astype(np.float32))
    v = torch.from_numpy(rng.randn(B, E, Tk).astype(np.float32))
    for la in (0, 5, 38):
        S = torch.bmm(q.transpose(1, 2), k)
        lo, hi = la - 1, la + 3
        if lo > 0:
            S[:, :, :lo] = -float("inf")
        if hi < Tk:
            S[:, :, hi:] = -float("inf")
        P = F.softmax(S, dim=-1)
        lat = torch.tensor([la], dtype=torch.int32, device=dev)
        _, Pg = ops.attention_core(q.to(dev), k.to(dev), v.to(dev), None, lat, win_back=1, win_ahead=3)
        assert rel_err(Pg.cpu(), P) < KTOL
        out = torch.zeros(1, dtype=torch.int32, device=dev)
        from deepvoice3_pytorch_amd import _lib
        _lib.call("dv3_attn_argmax_i32", Pg.data_ptr(), Tk, out.data_ptr(), ops._stream())
        assert int(out.item()) == int(P[0, 0].argmax())


def test_embedding_and_positions(dev):
    ops = _ops()
    rng = np.random.RandomState(4)
    B, T, C, V = 3, 45, 40, 30
    idx = torch.from_numpy(rng.randint(0, V, (B, T)))
    idx[:, -5:] = 0
    W = torch.from_numpy(rng.randn(V, C).astype(np.float32))
    wc = W.clone().requires_grad_(True)
    yc = F.embedding(idx, wc, padding_idx=0).transpose(1, 2)
    wg = W.clone().to(dev).requires_grad_(True)
    yg = ops.embedding_bct(idx.to(dev), wg, padding_idx=0)
    assert torch.equal(yg.detach().cpu(), yc.detach())
    wgt = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
    (gc,) = _grads((yc * wgt).sum(), [wc])
    (gg,) = _grads((yg * wgt.to(dev)).sum(), [wg])
    assert rel_err(gg.cpu(), gc) < 1e-6
    # sinusoidal encodings: scalar and per-batch rate, padding position 0
    table = O.position_encoding_table(64, C, 1.0, sinusoidal=False)
    pos = torch.from_numpy(rng.randint(0, 64, (B, T)))
    pos[:, -3:] = 0
    for w in (1.385, torch.tensor([0.7, 2.3, 7.6])):
        want = O.sinusoidal_encoding(table, pos, w).transpose(1, 2)
        got = ops.sincos_pos_bct(pos.to(dev), table.to(dev), w.to(dev) if torch.is_tensor(w) else w)
        assert np.abs(got.cpu().numpy() - want.numpy()).max() < 2e-5   # sin/cos of angles up to ~500 rad
    base = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
    got = ops.add_position_encoding(base.to(dev), pos.to(dev), table.to(dev), 1.0, True)
    assert np.abs(got.cpu().numpy() - (base + O.sinusoidal_encoding(table, pos, 1.0).transpose(1, 2)).numpy()).max() < 2e-5


def test_losses(dev):
    ops = _ops()
    from tests.util import load_golden
    fx = load_golden("losses")
    y_hat, y = torch.from_numpy(fx["spec/y_hat"]), torch.from_numpy(fx["spec/y"])
    lengths = torch.from_numpy(fx["spec/lengths"]).to(torch.int32)
    for wm, wbd in [(0.5, 0.1), (0.0, 0.1), (0.5, 0.0)]:
        tag = "spec_wm%g_wbd%g" % (wm, wbd)
        for layout in ("btc", "bct"):
            yh = y_hat.clone().to(dev)
            if layout == "bct":
                yh = yh.transpose(1, 2).contiguous().transpose(1, 2)
            yh.requires_grad_(True)
            out4 = ops.spec_loss(yh, y.to(dev), lengths.to(dev) if wm > 0 else None, 1, wm, wbd)
            out4[2].backward()
            assert rel_err(out4[0].detach().cpu(), fx[tag + "/l1"]) < 1e-5
            if wbd > 0:
                assert rel_err(out4[1].detach().cpu(), fx[tag + "/bd"]) < 1e-5
            assert rel_err(yh.grad.cpu(), fx[tag + "/grad"]) < 1e-5
    # guided attention: mean(attn * W) and its gradient W / N
    il, tl = fx["guided/in_len"], fx["guided/out_len"]
    rng = np.random.RandomState(1)
    attn = torch.from_numpy(rng.rand(2, 3, 12, 9).astype(np.float32))
    for g in (0.2, 0.4):
        W = torch.from_numpy(fx["guided_g%g" % g])
        a = attn.clone().to(dev).requires_grad_(True)
        loss = ops.guided_attention_loss(a, torch.from_numpy(il).to(torch.int32).to(dev),
                                         torch.from_numpy(tl).to(torch.int32).to(dev), g)
        loss.backward()
        assert rel_err(loss.detach().cpu(), (attn * W).mean()) < 1e-5
        assert rel_err(a.grad.cpu(), (W / attn.numel()).expand_as(attn)) < 1e-5
    p = torch.from_numpy(fx["bce/p"]).to(dev).requires_grad_(True)
    l = ops.bce_loss(p, torch.from_numpy(fx["bce/t"]).to(dev))
    l.backward()
    assert rel_err(l.detach().cpu(), fx["bce/loss"]) < 1e-5
    assert rel_err(p.grad.cpu(), fx["bce/grad"]) < 1e-5


def test_clip_adam(dev):
    ops = _ops()
    rng = np.random.RandomState(8)
    n = 100003
    p0 = torch.from_numpy(rng.randn(n).astype(np.float32))
    g0 = torch.from_numpy(rng.randn(n).astype(np.float32) * 0.01)
    pc, m, v = p0.clone(), torch.zeros(n), torch.zeros(n)
    pg, mg, vg = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    partial = torch.empty(1024, device=dev)
    out2 = torch.empty(2, device=dev)
    hyper = torch.empty(3, device=dev)
    for step in (1, 2, 3):
        grad = g0 * step
        gn = O.clip_and_adam([pc], [grad], [m], [v], step, 5e-4)
        gg = grad.clone().to(dev)
        ops.grad_sqnorm(gg, partial, out2)
        hyper.copy_(torch.tensor([5e-4, 1 - 0.5 ** step, math.sqrt(1 - 0.9 ** step)]))
        ops.clip_adam(pg, gg, mg, vg, out2, 0.1, hyper, 0.5, 0.9, 1e-6)
        assert rel_err(out2[0].cpu(), gn) < 1e-5
        assert rel_err(pg.cpu(), pc) < 1e-6
        assert rel_err(mg.cpu(), m) < 1e-5 and rel_err(vg.cpu(), v) < 1e-5


def test_conv_gemm_full_size_properties(dev):
    """BASELINE north-star shape (B=64, C=256, T=1024, k=3): too big for the CPU oracle in a unit
    test, so check size-independent properties: linearity in x of the pre-gate (a,b) outputs,
    shift-equivariance in time away from the borders, and a sampled direct dot-product check
    (fp64).  Runs on the kernel of the active GEMM mode."""
    ops = _ops()
    torch.manual_seed(0)
    B, C, T, k, d = 64, 256, 1024, 3, 3
    v = torch.randn(2 * C, C, k, device=dev) * 0.05
    g = torch.rand(2 * C, 1, 1, device=dev) + 0.5
    bias = torch.zeros(2 * C, device=dev)
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
    padL = d

    def pre_gate(x):
        ab = torch.empty(B, 2 * C, T, device=dev)
        ops.conv_gemm(x, pk.fwd, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL,
                      mode=ops.EPI_GLU, Cg=C, bias=bias, ab=ab, a_split=pk.fwd_s)   # fwd_s: None in f32 mode
        return ab
    x1, x2 = torch.randn(B, C, T, device=dev), torch.randn(B, C, T, device=dev)
    a1, a2, a12 = pre_gate(x1), pre_gate(x2), pre_gate(x1 + 2 * x2)
    assert rel_err((a1 + 2 * a2).cpu(), a12.cpu()) < (1e-5 if pk.fwd_s is None else 4e-5)
    xs = torch.roll(x1, 7, dims=2)
    a_s = pre_gate(xs)
    assert rel_err(a_s[:, :, 32:-32].cpu(), torch.roll(a1, 7, dims=2)[:, :, 32:-32].cpu()) < 1e-6
    # sampled direct check in float64
    w = (g * v / v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1)).double().cpu()
    xc = x1.double().cpu()
    rng = np.random.RandomState(0)
    for _ in range(64):
        b, o, t = rng.randint(B), rng.randint(2 * C), rng.randint(T)
        acc = 0.0
        for j in range(k):
            tt = t + j * d - padL
            if 0 <= tt < T:
                acc += float((w[o, :, j] * xc[b, :, tt]).sum())
        assert abs(acc - float(a1[b, o, t])) < (1e-5 if pk.fwd_s is None else 5e-5) * max(1.0, abs(acc))


@pytest.mark.parametrize("B,E,Tq,Tk,p", [(3, 48, 50, 37, 0.0), (2, 256, 201, 150, 0.05), (4, 20, 33, 64, 0.1),
                                         (1, 8, 2, 5, 0.0), (2, 64, 70, 511, 0.0)])
def test_fused_attention_forward_equals_unfused(dev, gemm_mode, B, E, Tq, Tk, p):
    """dv3_attn_fwd_f32 (scores -> mask -> softmax -> dropout -> context in one launch, exact fp32 MFMA) against the
    five-launch path it replaces and against the oracle arithmetic (deepvoice3.py:143-171); gradients flow through
    the unchanged backward from the P / pd it saves"""
    if gemm_mode != "f16x3":
        pytest.skip("the fused kernel is exact fp32 in every mode")
    ops = _ops()
    rng = np.random.RandomState(B + E + Tq + Tk)
    q = torch.from_numpy(rng.randn(B, E, Tq).astype(np.float32) * 0.3)
    k = torch.from_numpy(rng.randn(B, E, Tk).astype(np.float32) * 0.3)
    v = torch.from_numpy(rng.randn(B, E, Tk).astype(np.float32))
    key_len = torch.tensor([Tk] + [max(1, Tk - 3 * (i + 1)) for i in range(B - 1)], dtype=torch.int32)
    res = {}
    for fused in (False, True):
        ops.fused_attention = fused
        try:
            qg, kg, vg = [t.to(dev).requires_grad_(True) for t in (q, k, v)]
            ops.dropout_state.manual_seed(31)
            ctx, P = ops.attention_core(qg, kg, vg, key_len.to(dev), None, p, p > 0)
            w = torch.from_numpy(np.random.RandomState(1).randn(B, E, Tq).astype(np.float32)).to(dev)
            ((ctx * w).sum() + (P * P).sum()).backward()
            res[fused] = [t.detach().cpu() for t in (ctx, P, qg.grad, kg.grad, vg.grad)]
        finally:
            ops.fused_attention = True
    for a, b_, nm in zip(res[False], res[True], ("ctx", "P", "dq", "dk", "dv")):
        assert rel_err(b_, a) < 2e-5, nm
    if p == 0.0:      # against the reference arithmetic directly
        S = torch.bmm(q.transpose(1, 2), k)
        m = torch.arange(Tk)[None, None, :] >= key_len[:, None, None]
        Pw = torch.softmax(S.masked_fill(m, -float("inf")), dim=-1)
        want = torch.bmm(Pw, v.transpose(1, 2)).transpose(1, 2) * (Tk * math.sqrt(1.0 / Tk))
        assert rel_err(res[True][1], Pw) < 1e-5 and rel_err(res[True][0], want) < 1e-5


@pytest.mark.parametrize("mode", ["f16x3", "bf16x3"])
@pytest.mark.parametrize("B,M,C,T,d,masked,S", [(3, 128, 64, 75, 2, False, 2), (2, 512, 256, 150, 27, True, 3),
                                               (4, 96, 200, 61, 9, True, 5), (5, 256, 128, 800, 1, True, 7),
                                               (2, 130, 130, 33, 3, True, 2), (6, 72, 64, 100, 1, False, 19),
                                               (3, 128, 96, 50, -1, True, 3), (2, 256, 64, 804, -3, True, 4),
                                               (1, 70, 70, 9, 3, True, 1), (2, 128, 128, 16, -3, False, 2),
                                               (4, 512, 256, 201, 3, False, 6)])
def test_two_steps_ahead_wgrad_equals_the_all_taps_wgrad(dev, mode, B, M, C, T, d, masked, S):
    """wgrad_taps2_kernel (csrc/wgrad_taps2.hip, the default weight-gradient kernel: operand units of the NEXT two steps
    in flight as raw registers) against wgrad_taps_kernel: the K-slab partial sums are bit-identical (same tile, slab
    and accumulation order), ragged channel counts, dilations up to 27 and masked operands included; the variant each
    call served is read back (dv3_debug_get(11)).  Round 6: for d = 1 and 3 the default is the WINDOW form (one window of
    8 + 2 d elements per unit fetched, masked and converted once for the three taps: variant ...42): bit-identical to the
    per-tap form (dv3_debug_set(47, 0)) and to the all-taps kernel, same-padded (d > 0 here) and causal (d < 0: padL = 2 |d|),
    sequences shorter than a window included."""
    from deepvoice3_pytorch_amd import ops, _lib
    L = _lib.lib()
    causal, d = d < 0, abs(d)
    padL = 2 * d if causal else d
    prev = ops.set_gemm_precision(mode)
    try:
        torch.manual_seed(1)
        x = torch.randn(B, C, T, device=dev)
        g = torch.randn(B, M, T, device=dev)
        bits = rs = None
        if masked:
            ops.dropout_state.manual_seed(3)
            bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
        outs = []
        # dv3_debug_set(2, .): 3 = all-taps kernel, 4 = two-steps-ahead kernel; (47, .): its window form on / off
        for tile, win, il in ((3, 1, 1), (4, 1, 1), (4, 0, 0), (4, 1, 0)):
            L.dv3_debug_set(2, tile)
            L.dv3_debug_set(47, win)
            L.dv3_debug_set(48, il)         # the window form's staging between the MFMAs (default) / after them
            o = ops.wgrad_gemm(g, x, B=B, M=M, Cin=C, T=T, Tin=T, J=3, dil=d, padL=padL, n_slabs=S, xmask=bits,
                               xmask_rs=rs or 0, drop_scale=1 / 0.95 if masked else 1.0, split_bf16=True, k_split=True)
            outs.append((o.clone(), L.dv3_debug_get(11)))
    finally:
        L.dv3_debug_set(2, 0)
        L.dv3_debug_set(47, 1)
        L.dv3_debug_set(48, 1)
        ops.set_gemm_precision(prev)
    windowed = d in (1, 3) and (B - 1) * C * T + (C - 1) * T + T >= 16
    assert [o[1] % 1000 for o in outs] == [30, 46 if windowed else 40, 40, 42 if windowed else 40], [o[1] for o in outs]
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]), (o[1], float((outs[0][0] - o[0]).abs().max()))


@pytest.mark.parametrize("B,C,T,d,causal,masked", [(3, 64, 75, 2, False, True), (2, 256, 150, 27, False, False),
                                                   (2, 128, 100, 1, True, True), (5, 96, 61, 9, False, True),
                                                   (4, 256, 800, 3, False, True), (7, 32, 33, 1, False, False)])
def test_256x256_k16_pingpong_tap_gemm_equals_the_128_wide_kernels(dev, gemm_mode, B, C, T, d, causal, masked):
    """csrc/conv_gemm_pp2.hip (tile_hint 30; the picker's choice from 128 tiles up): same MFMAs per accumulator in the
    same order as the 128-row kernels -- forward (Conv1dGLU, modules.py:145-164, with the pre-gate save, dropout as
    keep-bytes of the same decisions) and input-gradient form agree bit for bit in the default mode; in bf16x3 the
    masked forward differs in the last bit of some lo operands (the mask multiply is contracted into the residual in
    one kernel and not the other) and is held to the kernel tolerance."""
    if gemm_mode == "f32":
        pytest.skip("split-operand kernel test")
    from deepvoice3_pytorch_amd import ops, _lib
    k = 3
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.randn(2 * C, device=dev) * 0.1
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True)
    bits = rs = kb = None
    if masked:
        ops.dropout_state.manual_seed(3)
        bits, rs, kb = ops.dropout_bits_keep(B, C, T, 0.05, dev)
        ops.dropout_state.manual_seed(3)
        bits2, _ = ops.dropout_bits(B * C, T, 0.05, dev)
        assert torch.equal(bits, bits2) and torch.equal(kb, ops.mask_bits_to_c8(bits2, rs, B, C, T))
    padL = (k - 1) * d if causal else d
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
              residual=1, a_split=pk.fwd_s, xmask=bits, xmask_rs=rs or 0, xmask_c8=kb,
              drop_scale=1 / 0.95 if masked else 1.0)
    ys, abs_ = [], []
    for hint in (21 if C >= 64 else 0, 30):
        y = torch.empty(B, C, T, device=dev)
        ab = torch.empty(B, 2 * C, T, device=dev)
        ops.conv_gemm(x, None, pk.lda, pk.a_half, y=y, ab=ab, tile_hint=hint, **kw)
        ys.append(y)
        abs_.append(ab)
    assert _lib.lib().dv3_debug_get(10) % 1000 == 101
    if gemm_mode == "f16x3" or not masked:
        assert torch.equal(ys[0], ys[1]) and torch.equal(abs_[0], abs_[1])
    else:
        assert rel_err(ys[1].cpu(), ys[0].cpu()) < KTOL and rel_err(abs_[1].cpu(), abs_[0].cpu()) < KTOL
    gm = torch.randn(B, 2 * C, T, device=dev)
    dres = torch.randn(B, C, T, device=dev)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=d, padL=(k - 1) * d - padL, mode=ops.EPI_DGRAD,
               r=dres, ymask=bits, ymask_rs=rs or 0, drop_scale=1 / 0.95 if masked else 1.0, a_split=pk.bwd_s)
    dxs = []
    try:
        # (the picker's kernel for these small grids is the k-split form since round 5 -- first half + second half of the
        #  chunk range: compared with the running sum of the 256 x 256 kernel it is held to the tolerance, the one-group
        #  loop bit for bit)
        for hint, ks in ((0, 0), (30, 0), (0, KS_DEFAULT)):
            _lib.lib().dv3_debug_set(44, ks)
            dx = torch.empty(B, C, T, device=dev)
            ops.conv_gemm(gm, None, pk.ldb, 0, y=dx, tile_hint=hint, **dkw)
            dxs.append(dx)
    finally:
        _lib.lib().dv3_debug_set(44, KS_DEFAULT)
    assert torch.equal(dxs[0], dxs[1])
    assert rel_err(dxs[2].cpu(), dxs[1].cpu()) < (5e-6 if gemm_mode == "f16x3" else KTOL)


@pytest.mark.parametrize("B,C,T,k,d,causal,masked", [(3, 64, 75, 3, 2, False, True), (2, 256, 150, 3, 27, False, False),
                                                     (2, 128, 100, 3, 1, True, True), (5, 96, 61, 3, 9, False, True),
                                                     (7, 32, 33, 3, 1, False, False), (4, 160, 201, 1, 1, False, False),
                                                     (3, 24, 37, 1, 1, False, True), (2, 512, 150, 3, 3, False, True),
                                                     (16, 256, 201, 1, 1, False, True), (1, 40, 17, 3, 27, True, False)])
def test_deep_prefetch_form_of_the_128x64_tile_is_bit_identical(dev, gemm_mode, B, C, T, k, d, causal, masked):
    """conv_gemm_bf16x3.hip, template DPJ (round 5; experiment build, dv3_debug_set(43, 0 | 1 | 2)): the 128 x 64 split tile with its
    global fetches 3-4 steps ahead through register rings -- same fragment images, same MFMA order, so forward
    (Conv1dGLU with the pre-gate save and keep-bits, modules.py:145-164), plain 1 x 1 / Linear and the input-gradient
    form must agree BIT FOR BIT with the in-phase loop; covers 1 and 3 taps, partial chunks (C % 32 != 0), tiles shorter
    than the rings (C = 24, 32, 40: one or two steps), columns across batch items, and the fp16 range guard staying
    silent on the rings' virtual steps."""
    if gemm_mode == "f32":
        pytest.skip("split-operand kernel test")
    from deepvoice3_pytorch_amd import ops, _lib
    L = _lib.lib()
    if L.dv3_debug_set(43, 2) != 0:
        pytest.skip("the deep-prefetch form was measured and retired (profiles/r05_deep_prefetch_rings.txt): it is compiled "
                    "into the experiment build only (make EXP=1, DV3_LIBPATH=.../libdv3hip_exp.so)")
    L.dv3_debug_set(43, 0)
    L.dv3_debug_set(44, 0)          # (the rule would give these small grids to the k-split form)
    torch.manual_seed(C + T)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.randn(2 * C, device=dev) * 0.1
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True)
    bits = rs = None
    if masked:
        ops.dropout_state.manual_seed(3)
        bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
    padL = (k - 1) * d if causal else (k - 1) // 2 * d
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
              residual=1, a_split=pk.fwd_s, xmask=bits, xmask_rs=rs or 0, drop_scale=1 / 0.95 if masked else 1.0,
              tile_hint=22)
    gm = torch.randn(B, 2 * C, T, device=dev)
    dres = torch.randn(B, C, T, device=dev)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=d, padL=(k - 1) * d - padL, mode=ops.EPI_DGRAD,
               r=dres, ymask=bits, ymask_rs=rs or 0, drop_scale=1 / 0.95 if masked else 1.0, a_split=pk.bwd_s, tile_hint=22)
    lkw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL, mode=ops.EPI_RELU, bias=bias, a_split=pk.fwd_s,
               tile_hint=22)
    ev0 = ops.f16_range_events() if hasattr(ops, "f16_range_events") else None
    outs = []
    try:
        for dp in (0, 2):
            L.dv3_debug_set(43, dp)
            y = torch.empty(B, C, T, device=dev)
            ab = torch.empty(B, 2 * C, T, device=dev)
            dx = torch.empty(B, C, T, device=dev)
            z = torch.empty(B, 2 * C, T, device=dev)
            ops.conv_gemm(x, None, pk.lda, pk.a_half, y=y, ab=ab, **kw)
            v0 = L.dv3_debug_get(10)
            ops.conv_gemm(gm, None, pk.ldb, 0, y=dx, **dkw)
            v1 = L.dv3_debug_get(10)
            ops.conv_gemm(x, None, pk.lda, pk.lda, y=z, **lkw)
            v2 = L.dv3_debug_get(10)
            assert (v0 % 10, v1 % 10, v2 % 10) == ((5, 5, 5) if dp else (0, 0, 0)), (dp, v0, v1, v2)
            assert v0 % 1000 // 10 == 2 and v1 % 1000 // 10 == 2
            outs.append((y, ab, dx, z))
    finally:
        L.dv3_debug_set(43, DP_DEFAULT)
        L.dv3_debug_set(44, KS_DEFAULT)
    for a, b, name in zip(outs[0], outs[1], ("y", "pre-gate", "dx", "relu")):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (name, float((a - b).abs().max()))
    if ev0 is not None:
        assert ops.f16_range_events() == ev0


@pytest.mark.parametrize("B,C,T,k,d,causal,masked", [(3, 64, 75, 3, 2, False, True), (2, 256, 150, 3, 27, False, False),
                                                     (2, 128, 100, 3, 1, True, True), (5, 96, 61, 3, 9, False, True),
                                                     (4, 160, 201, 1, 1, False, False), (3, 40, 37, 1, 1, False, True),
                                                     (2, 512, 150, 3, 3, False, True), (16, 256, 201, 1, 1, False, True),
                                                     (1, 72, 17, 3, 27, True, False), (3, 24, 50, 3, 1, False, False)])
def test_k_split_form_of_the_128x64_tile(dev, gemm_mode, B, C, T, k, d, causal, masked):
    """conv_gemm_bf16x3.hip, template KS = 2 (round 5; dv3_debug_set(44, 0 | 1 | 2)): two wave groups of one workgroup
    take the first / second half of the input-channel chunks of the same 128 x 64 tile and the second hands its
    accumulators to the first through LDS.  The sum is (first half) + (second half) instead of one running sum:
    not bit-identical to the one-group loop, but a fixed function of the shape -- checked: equal to the one-group
    result to the kernel tolerance, REPEATABLE bit for bit, and anchored on the oracle (Conv1dGLU forward,
    modules.py:145-164).  Covers odd chunk counts (C = 96, 160: the second group runs a chunk fewer), a partial last
    chunk in the second group (C = 40, 72), 1 and 3 taps, keep-bits, the input-gradient form, and a single-chunk
    layer (C = 24) that must stay on the one-group kernel."""
    if gemm_mode == "f32":
        pytest.skip("split-operand kernel test")
    from deepvoice3_pytorch_amd import ops, _lib
    L = _lib.lib()
    rng = np.random.RandomState(C + T + k)
    sd = _glu_sd(C, k, rng)
    x_cpu = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
    x = x_cpu.to(dev)
    bias = sd["l.conv.bias"].to(dev)
    pk = ops.pack_weights(sd["l.conv.weight_v"].to(dev), sd["l.conv.weight_g"].to(dev), glu_cg=C, need_bwd=True)
    bits = rs = None
    if masked:
        ops.dropout_state.manual_seed(3)
        bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
    padL = (k - 1) * d if causal else (k - 1) // 2 * d
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
              residual=1, a_split=pk.fwd_s, xmask=bits, xmask_rs=rs or 0, drop_scale=1 / 0.95 if masked else 1.0,
              tile_hint=22)
    torch.manual_seed(1)
    gm = torch.randn(B, 2 * C, T, device=dev)
    dres = torch.randn(B, C, T, device=dev)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=d, padL=(k - 1) * d - padL, mode=ops.EPI_DGRAD,
               r=dres, ymask=bits, ymask_rs=rs or 0, drop_scale=1 / 0.95 if masked else 1.0, a_split=pk.bwd_s, tile_hint=22)
    eligible = (C + 31) // 32 >= 2
    outs = []
    try:
        for ks in (0, 2, 2):
            L.dv3_debug_set(44, ks)
            y = torch.empty(B, C, T, device=dev)
            ab = torch.empty(B, 2 * C, T, device=dev)
            dx = torch.empty(B, C, T, device=dev)
            ops.conv_gemm(x, None, pk.lda, pk.a_half, y=y, ab=ab, **kw)
            v0 = L.dv3_debug_get(10)
            ops.conv_gemm(gm, None, pk.ldb, 0, y=dx, **dkw)
            v1 = L.dv3_debug_get(10)
            assert (v0 % 10, v1 % 10) == ((2, 2) if ks and eligible else (0, 2 if ks else 0)), (ks, v0, v1)
            assert v0 % 1000 // 10 == 2 and v1 % 1000 // 10 == 2
            outs.append((y, ab, dx))
    finally:
        L.dv3_debug_set(44, KS_DEFAULT)
    tol = 5e-6 if gemm_mode == "f16x3" else KTOL
    for a, b, c, name in zip(outs[0], outs[1], outs[2], ("y", "pre-gate", "dx")):
        assert torch.equal(b.view(torch.int32), c.view(torch.int32)), name          # repeatable
        assert rel_err(b.cpu(), a.cpu()) < tol, (name, rel_err(b.cpu(), a.cpu()))
    if not masked:
        assert rel_err(outs[1][0].cpu(), O.conv1d_glu(sd, "l", x_cpu, k, d, causal, True)) < KTOL


def test_stream_k_workspace_never_comes_from_a_capture_pool(dev):
    """ops._streamk_ws (round 5's NaN: the nyanko bf16 replay after deepvoice3 f16x3 replays in one process):
    the workspace of a stream is allocated eagerly from the ordinary pool.  A launch captured on a stream that has none
    gets NO workspace (tile-per-workgroup form, variant ...101) and caches nothing; once its owner has prepared one
    (ops.prepare_streamk_ws, what GraphedTrainer does before it captures) the captured launch takes the stream-K form
    (...102), replays give the eager result, and the buffer outlives the graph."""
    from deepvoice3_pytorch_amd import ops, _lib
    L = _lib.lib()
    ops.set_gemm_precision("f16x3")
    B, C, T, k = 64, 512, 150, 3                     # the encoder layer of the benchmark step: 152 tiles, stream-K by the rule
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C, r=x, residual=1, a_split=pk.fwd_s)
    y_eager = torch.empty(B, C, T, device=dev)
    ops.conv_gemm(x, None, pk.lda, pk.a_half, y=y_eager, **kw)
    assert L.dv3_debug_get(10) % 1000 == 102            # eager on the current stream: stream-K form
    torch.cuda.synchronize()
    for prepared in (False, True):
        cap = torch.cuda.Stream()
        while (dev.index or 0, cap.cuda_stream) in ops._sk_ws:     # a pool stream nobody has used for this yet
            cap = torch.cuda.Stream()
        n_keys = len(ops._sk_ws)
        if prepared:
            assert ops.prepare_streamk_ws(dev, cap) is not None and len(ops._sk_ws) == n_keys + 1
        y = torch.empty(B, C, T, device=dev)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=cap):
            ops.conv_gemm(x, None, pk.lda, pk.a_half, y=y, **kw)
            variant = L.dv3_debug_get(10) % 1000
        assert variant == (102 if prepared else 101), (prepared, variant)
        assert len(ops._sk_ws) == n_keys + (1 if prepared else 0)
        for _ in range(3):
            y.zero_()
            gr.replay()
            torch.cuda.synchronize()
            if prepared:
                assert torch.equal(y, y_eager)
            else:
                assert float((y - y_eager).abs().max()) < 2e-6 * float(y_eager.abs().max())
        ptr = ops._sk_ws.get((dev.index or 0, cap.cuda_stream), (None,))[0]
        del gr
        torch.cuda.synchronize()
        if prepared:      # still alive and still zeroed flags after the graph is gone
            e = ops._sk_ws[(dev.index or 0, cap.cuda_stream)]
            assert e[0] == ptr and int(e[2][:1024].abs().sum()) == 0



@pytest.mark.parametrize("B,C,T,d,masked", [(64, 512, 150, 1, True), (64, 512, 150, 27, False), (48, 256, 201, 3, True),
                                            (24, 256, 410, 1, False)])
def test_stream_k_form_of_the_256x256_tap_gemm(dev, gemm_mode, B, C, T, d, masked):
    """csrc/conv_gemm_pp2.hip, stream-K form (one workgroup per CU over equal shares of (tile, 32-channel chunk) units, a cut
    tile summed by the workgroup that holds its first chunk): same operands, same products, fp32 partial sums added in
    another order -- within 2e-6 of the tile-per-workgroup form relative to the output's range, run-to-run bit identical
    (the cut is a function of the shape), flags left zero for the next launch (three launches on one workspace), forward
    (Conv1dGLU with the pre-gate save, dropout as keep-bytes; modules.py:145-164) and input-gradient form.  The first
    shape is the encoder layer of the benchmark step (152 tiles on 256 CUs), the last a grid smaller than the chip."""
    if gemm_mode == "f32":
        pytest.skip("split-operand kernel test")
    from deepvoice3_pytorch_amd import ops, _lib
    L = _lib.lib()
    k = 3
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.randn(2 * C, device=dev) * 0.1
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True)
    bits = rs = kb = None
    if masked:
        ops.dropout_state.manual_seed(3)
        bits, rs, kb = ops.dropout_bits_keep(B, C, T, 0.05, dev)
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=d, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x, residual=1,
              a_split=pk.fwd_s, xmask=bits, xmask_rs=rs or 0, xmask_c8=kb, drop_scale=1 / 0.95 if masked else 1.0, tile_hint=30)
    gm = torch.randn(B, 2 * C, T, device=dev)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=d, padL=(k - 1) * d - d, mode=ops.EPI_DGRAD, r=x, r_scale=0.7071,
               ymask=bits, ymask_rs=rs or 0, drop_scale=1 / 0.95 if masked else 1.0, a_split=pk.bwd_s, tile_hint=30)
    prev = ops.streamk
    ops.streamk = "force"
    try:
        res = {}
        for sk in (0, 2, 2, 2):
            L.dv3_debug_set(22, sk)
            y, ab, dx = torch.empty(B, C, T, device=dev), torch.empty(B, 2 * C, T, device=dev), torch.empty(B, C, T, device=dev)
            ops.conv_gemm(x, None, pk.lda, pk.a_half, y=y, ab=ab, **kw)
            vf = L.dv3_debug_get(10)
            ops.conv_gemm(gm, None, pk.ldb, 0, y=dx, **dkw)
            vd = L.dv3_debug_get(10)
            res.setdefault(sk, []).append((y, ab, dx, vf % 1000, vd % 1000))
    finally:
        L.dv3_debug_set(22, 1)
        ops.streamk = prev
    torch.cuda.synchronize()
    (y0, ab0, dx0, vf0, vd0), sks = res[0][0], res[2]
    assert vf0 == 101 and vd0 == 101 and sks[0][3] == 102 and sks[0][4] == 102
    for (y1, ab1, dx1, _, _) in sks[1:]:
        assert torch.equal(y1, sks[0][0]) and torch.equal(ab1, sks[0][1]) and torch.equal(dx1, sks[0][2])
    for a, b in ((y0, sks[0][0]), (ab0, sks[0][1]), (dx0, sks[0][2])):
        assert float((a - b).abs().max()) < 2e-6 * float(a.abs().max())


@pytest.mark.parametrize("O_,I,J,S,rows", [(512, 256, 3, 14, True), (1024, 512, 3, 32, True), (256, 256, 1, 5, True),
                                           (96, 64, 5, 9, False), (513, 512, 1, 3, True), (64, 36, 3, 17, True)])
def test_weight_norm_backward_16_byte_gather_is_the_4_byte_one(dev, O_, I, J, S, rows):
    """dv3_weight_norm_bwd_f32 sums the K-split partial slabs four columns per thread (round 6) with the scalar loop's
    order of additions per element: dv, dg and dbias must be the same bits as with dv3_debug_set(51, 0)."""
    from deepvoice3_pytorch_amd import ops, _lib
    rng = np.random.RandomState(O_ + I + J + S)
    ldo = I
    slabs = torch.from_numpy(rng.randn(*((J, O_, S, ldo) if rows else (S, J, O_, ldo))).astype(np.float32)).to(dev)
    v = torch.from_numpy(rng.randn(O_, I, J).astype(np.float32) * 0.1).to(dev)
    g = torch.from_numpy(rng.uniform(0.5, 1.5, (O_, 1, 1)).astype(np.float32)).to(dev)
    scale = 1.0 / v.reshape(O_, -1).norm(dim=1)
    part = torch.from_numpy(rng.randn(7, O_).astype(np.float32)).to(dev)
    outs = []
    for sw in (0, 1):
        _lib.lib().dv3_debug_set(51, sw)
        try:
            outs.append(ops.weight_norm_bwd(slabs, S, ldo, v, g, scale, part, 7, O_, I, J, rows_of_slabs=rows))
        finally:
            _lib.lib().dv3_debug_set(51, 1)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # and it is the weight-norm backward: dW = sum of the slabs, dg = <dW, v> / |v|, dv = g / |v| (dW - <dW, v> v / |v|^2)
    dW = (slabs.sum(2) if rows else slabs.sum(0)).permute(1, 2, 0).double()          # [O][I][J]
    vd = v.double()
    nrm = vd.reshape(O_, -1).norm(dim=1).view(O_, 1, 1)
    dot = (dW * vd).reshape(O_, -1).sum(1).view(O_, 1, 1)
    dg_ref = dot / nrm
    dv_ref = g.double() / nrm * (dW - dot * vd / nrm ** 2)
    dv, dg, dbias = outs[1]
    assert rel_err(dv.cpu().numpy(), dv_ref.cpu().numpy()) < 2e-5
    assert rel_err(dg.cpu().numpy(), dg_ref.cpu().numpy()) < 2e-5
    assert rel_err(dbias.cpu().numpy(), part.double().sum(0).cpu().numpy()) < 2e-5


@pytest.mark.parametrize("mode,B,C,T", [("glu", 3, 64, 201), ("glu", 2, 128, 150), ("glu", 2, 8, 3), ("glu", 5, 4, 1),
                                       ("glu16", 3, 64, 203), ("highway", 2, 96, 67), ("relu", 3, 65, 201),
                                       ("sigmoid", 2, 513, 81), ("linear", 3, 80, 7), ("softsign", 2, 33, 2),
                                       ("glu", 2, 64, 64)])
@pytest.mark.parametrize("pair", [False, True])
def test_gate_backward_16_byte_rows_of_any_length(dev, mode, B, C, T, pair):
    """dv3_gate_bwd_f32 reads and writes 16 bytes per lane for ANY T (head / aligned quads / tail of a row whose start is
    only 4-byte aligned; round 6, default) -- element gradients must be the bits of the 4-byte form (dv3_debug_set(55, 0)),
    the row sums differ by the order of their additions only, and both are the autograd of modules.py:157-164 / :224-226."""
    from deepvoice3_pytorch_amd import ops, _lib
    gated = mode in ("glu", "glu16", "highway")
    if pair and not gated:
        pytest.skip("pair words are the gated layers' pre-gate gradient")
    M = {"glu": ops.EPI_GLU, "glu16": ops.EPI_GLU, "highway": ops.EPI_HIGHWAY, "relu": ops.EPI_RELU,
         "sigmoid": ops.EPI_SIGMOID, "linear": ops.EPI_LINEAR, "softsign": ops.EPI_SOFTSIGN}[mode]
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + C + T)
    dy = torch.randn(B, C, T, generator=g).to(dev)
    ab = torch.randn(B, 2 * C if gated else C, T, generator=g).to(dev)
    if mode == "sigmoid":
        ab = torch.sigmoid(ab)
    if mode == "softsign":
        ab = ab / (1 + ab.abs())
    if mode == "glu16":
        ab = ab.to(torch.bfloat16)
    x = torch.randn(B, C, T, generator=g).to(dev) if mode == "highway" else None
    res = []
    for sw in (0, 1):
        _lib.lib().dv3_debug_set(55, sw)
        try:
            res.append(ops.gate_bwd(dy, None if mode == "linear" else ab, x, B=B, C=C, T=T, mode=M,
                                    residual=1 if mode.startswith("glu") else 0, pair=pair, want_dres=mode == "highway",
                                    alpha=1.0 if gated else 0.7))
        finally:
            _lib.lib().dv3_debug_set(55, 1)
    for a, b in zip(res[0][:2], res[1][:2]):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert float((res[0][2] - res[1][2]).abs().max()) <= 4e-6 * max(1.0, float(res[0][2].abs().max()))
    # against autograd in double precision
    dab, dres, part = res[1]
    if pair:
        w = dab.view(torch.int32)
        dab = ((w & -65536).view(torch.float32) + (w << 16).view(torch.float32))
    abd, dyd = ab.double(), dy.double()
    if gated:
        a, gt = abd[:, :C], abd[:, C:]
        s = torch.sigmoid(gt)
        if mode == "highway":
            d = dyd
            want = torch.cat([d * s, d * (a - x.double()) * s * (1 - s)], 1)
            assert rel_err(dres.cpu().numpy(), (d * (1 - s)).cpu().numpy()) < 1e-5
        else:
            d = dyd * math.sqrt(0.5)
            want = torch.cat([d * s, d * a * s * (1 - s)], 1)
    else:
        d = dyd * 0.7
        want = {"relu": d * (abd > 0), "sigmoid": d * abd * (1 - abd), "linear": d,
                "softsign": d * (1 - abd.abs()) ** 2}[mode]
    assert rel_err(dab.cpu().numpy(), want.cpu().numpy()) < (3e-5 if pair else 1e-5)
    assert rel_err(part.cpu().numpy(), want.sum(2).cpu().numpy()) < 2e-5


def test_memset_is_a_fill_kernel_with_exact_extent(dev):
    """dv3_memset_b8 (ops.zero_: the gradient arena, padded c8 tensors) fills exactly [p, p + bytes) for every alignment of
    both ends -- it is a kernel since round 6 (the runtime's memset node replayed with a corrupt pattern inside a captured
    step: profiles/r06_memset_node.txt)"""
    from deepvoice3_pytorch_amd import ops, _lib
    buf = torch.empty(1 << 16, dtype=torch.uint8, device=dev)
    for start, n, val in [(0, 65536, 0), (1, 17, 7), (3, 40000, 255), (16, 16, 1), (15, 1, 9), (5, 0, 3), (32, 4097, 0), (7, 31, 5)]:
        buf.fill_(0xAB)
        _lib.call("dv3_memset_b8", buf.data_ptr() + start, val, n, ops._stream())
        want = torch.full_like(buf, 0xAB)
        want[start:start + n] = val
        assert torch.equal(buf, want), (start, n, val)
    big = torch.full((25_000_001,), 3.0, device=dev)
    ops.zero_(big)
    assert float(big.abs().max()) == 0.0
    # dv3_memset_rows_b8: runs a stride apart -- the padding groups of a c8 tensor (ops._c8_empty: 513 channels -> groups 64..67)
    buf.fill_(0xAB)
    _lib.call("dv3_memset_rows_b8", buf.data_ptr() + 64, 0, 5, 48, 1024, ops._stream())
    want = torch.full_like(buf, 0xAB)
    for r in range(5):
        want[64 + r * 1024:64 + r * 1024 + 48] = 0
    assert torch.equal(buf, want)
    t = ops._c8_empty(3, 513, 77, dev)
    assert t.shape == (3, 68, 77, 8) and float(t[:, 64:].float().abs().max()) == 0.0
