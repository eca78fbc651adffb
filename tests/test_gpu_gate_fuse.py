# coding: utf-8
"""Round 6: the gate backward of a Conv1dGLU / HighwayConv1d inside the input-gradient launch of its consumer
(ops.GateFuse; include/dv3hip.h: dv3_conv_desc.pg / pg_pair / x_pair, dv3_wgrad_desc.g_pair).

Reference semantics: autograd of deepvoice3_pytorch/modules.py:157-164 (GLU) and :224-226 (highway).  Two checks per
chain `producer (gated) -> consumer (any conv-like layer)`:
  * against the CPU oracle's autograd (the tolerance of tests/test_gpu_kernels.py::test_conv_layer_backward);
  * against the same chain with the stand-alone dv3_gate_bwd_f32 launches (ops.fuse_gate_bwd = False): the input gradient
    and the weight gradients must agree BIT FOR BIT -- the tail runs the function the stand-alone kernel runs
    (common.h: dv3_gate_deriv) and the pair words are the operands the gradient GEMMs would have built themselves; the
    bias gradients differ by the order of their partial sums only.
The kernel variant that served each launch is asserted (dv3_debug_get(10 / 11)).
"""
import numpy as np
import pytest
import torch

from oracle import dv3_oracle as O
from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True, params=["f16x3", "bf16x3"])
def gemm_mode(request):
    from deepvoice3_pytorch_amd import ops
    prev = ops.set_gemm_precision(request.param)
    prev_max, ops.fuse_gate_max_elems = ops.fuse_gate_max_elems, 1 << 40      # the size rule is a speed rule: off here
    yield request.param
    ops.fuse_gate_max_elems = prev_max
    ops.set_gemm_precision(prev)


def _gated_params(rng, C, k):
    v = torch.from_numpy((rng.randn(2 * C, C, k) * (0.6 / np.sqrt(C * k))).astype(np.float32))
    g = torch.from_numpy(rng.uniform(0.5, 1.5, (2 * C, 1, 1)).astype(np.float32))
    b = torch.from_numpy(rng.uniform(-0.2, 0.2, 2 * C).astype(np.float32))
    return v, g, b


def _consumer(ops, rng, kind, C, k, d):
    """-> (params, cfg, cpu_forward(sd-style))"""
    if kind in ("glu", "highway"):
        v, g, b = _gated_params(rng, C, k)
        cfg = ops.LayerCfg(k=k, dil=d, causal=(kind == "highway"), mode=ops.EPI_GLU if kind == "glu" else ops.EPI_HIGHWAY,
                           residual=(kind == "glu"), p=0.1, training=True, site="cons")
    elif kind == "convT":
        v = torch.from_numpy(rng.randn(C, 24, 2).astype(np.float32) * 0.1)
        g = torch.from_numpy(rng.uniform(0.5, 1.5, (C, 1, 1)).astype(np.float32))
        b = torch.from_numpy(rng.uniform(-0.2, 0.2, 24).astype(np.float32))
        cfg = ops.LayerCfg(k=2, mode=ops.EPI_LINEAR, transposed=True)
    else:
        Co = 72 if C < 200 else C + 1
        v = torch.from_numpy(rng.randn(Co, C, 1).astype(np.float32) * 0.1)
        g = torch.from_numpy(rng.uniform(0.5, 1.5, (Co, 1, 1)).astype(np.float32))
        b = torch.from_numpy(rng.uniform(-0.2, 0.2, Co).astype(np.float32))
        cfg = ops.LayerCfg(mode=ops.EPI_RELU if kind == "relu1x1" else ops.EPI_LINEAR)
    return (v, g, b), cfg


def _run_chain(ops, dev, tensors, pcfg, ccfg, wgt, fuse, seed=123):
    """producer -> consumer on the GPU; returns (y, grads of [x, pv, pg, pb, cv, cg, cb], masks record, variants)"""
    ops.dropout_state.manual_seed(seed)
    ops.dropout_state.record = {}
    prev = ops.fuse_gate_bwd
    ops.fuse_gate_bwd = fuse
    try:
        gin = [t.clone().to(dev).requires_grad_(True) for t in tensors]
        x, pv, pg, pb, cv, cg, cb = gin
        h = ops.conv_layer(x, pv, pg, pb, pcfg)
        if fuse:
            ops.mark_sole_consumer(h)
            assert getattr(h, "_dv3_sole", False), "the producer left no token on its output"
        y = ops.conv_layer(h, cv, cg, cb, ccfg)
        before = dict(ops.gate_fuse_stats)
        grads = torch.autograd.grad((y * wgt.to(dev)).sum(), gin)
        torch.cuda.synchronize()
        stats = {k: ops.gate_fuse_stats[k] - before[k] for k in before}
        rec = {k: (b.cpu().numpy().view(np.uint32).copy(), rows, TT) for k, (b, rows, TT) in ops.dropout_state.record.items()}
        return y.detach().cpu(), [t.cpu() for t in grads], rec, stats
    finally:
        ops.fuse_gate_bwd = prev
        ops.dropout_state.record = None


CHAINS = [
    # producer kind, consumer kind, B, C, T, k, d          what it exercises
    ("glu_res", "glu", 3, 64, 150, 3, 3),        # narrow tail (T % 4 != 0), 128 x 64 tiles
    ("glu_res", "glu", 2, 64, 804, 3, 1),        # wide tail, partial last 32-column block
    ("glu", "linear", 3, 40, 61, 3, 1),          # M and Cin <= 64: fp32 pre-gate gradient (no pair words), odd sizes
    ("glu_res", "relu1x1", 2, 256, 200, 3, 27),  # the consumer is a 1 x 1 conv + ReLU (its own gate_bwd stays)
    ("glu_res", "convT", 2, 64, 100, 3, 1),      # ConvTranspose1d consumer (M = 2 O rows, J = 1)
    ("highway", "highway", 3, 64, 152, 3, 9),    # highway producer (dres written), causal
    ("highway", "glu", 2, 96, 75, 3, 1),
    ("glu_res", "glu", 4, 512, 1024, 3, 1),      # 256 x 256 kernel (128 tiles), pair-word staging
    ("glu_res", "glu", 8, 256, 804, 3, 3),       # 256 x 256 kernel on a ragged column count, wide tail
]


@pytest.mark.parametrize("pkind,ckind,B,C,T,k,d", CHAINS)
def test_fused_gate_backward_matches_standalone_and_oracle(dev, gemm_mode, pkind, ckind, B, C, T, k, d):
    from deepvoice3_pytorch_amd import ops, _lib
    rng = np.random.RandomState(B * 1000 + C + T)
    p = 0.1
    pv, pg, pb = _gated_params(rng, C, k)
    pcfg = ops.LayerCfg(k=k, dil=d, causal=(pkind == "highway"), mode=ops.EPI_HIGHWAY if pkind == "highway" else ops.EPI_GLU,
                        residual=(pkind == "glu_res"), p=p, training=True, site="prod")
    (cv, cg, cb), ccfg = _consumer(ops, rng, ckind, C, k, d)
    x = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
    tensors = [x, pv, pg, pb, cv, cg, cb]
    To = 2 * T if ckind == "convT" else T
    Co = {"glu": C, "highway": C, "convT": 24}.get(ckind, cv.shape[0])
    wgt = torch.from_numpy(rng.randn(B, Co, To).astype(np.float32))

    y1, g1, rec, st1 = _run_chain(ops, dev, tensors, pcfg, ccfg, wgt, True)
    y0, g0, _, st0 = _run_chain(ops, dev, tensors, pcfg, ccfg, wgt, False)
    assert st1["fused"] == 1, st1                       # the producer's backward took the fused result
    assert st0["fused"] == 0 and st0["standalone"] >= 1, st0
    assert torch.equal(y1, y0)
    names = ("dx", "p.dv", "p.dg", "p.dbias", "c.dv", "c.dg", "c.dbias")
    for n, a, b in zip(names, g1, g0):
        if n == "p.dbias":                                # 32-column partial sums vs per-batch-item partial sums
            assert rel_err(a, b) < 2e-6, n
        else:
            assert torch.equal(a, b), "%s differs from the stand-alone gate backward: %g" % (n, rel_err(a, b))

    # ---- the oracle's autograd on the same masks ----
    def keep_of(site):
        bits, rows, TT = rec[site]
        return torch.from_numpy(O.unpack_keep_bits(bits, rows, (TT + 31) // 32, TT)).view(B, C, TT).float()

    def drop(site, t, pp, layout):
        return t * keep_of({"p": "prod", "c": "cons"}[site]) / (1 - pp)
    cin = [t.clone().requires_grad_(True) for t in tensors]
    sd = {"p.conv.weight_v": cin[1], "p.conv.weight_g": cin[2], "p.conv.bias": cin[3],
          "c.conv.weight_v": cin[4], "c.conv.weight_g": cin[5], "c.conv.bias": cin[6],
          "c.weight_v": cin[4], "c.weight_g": cin[5], "c.bias": cin[6]}
    if pkind == "highway":
        h = O.highway_conv1d(sd, "p", cin[0], k, d, True, p, drop)
    else:
        h = O.conv1d_glu(sd, "p", cin[0], k, d, False, pkind == "glu_res", p, drop)
    if ckind == "glu":
        yc = O.conv1d_glu(sd, "c", h, k, d, False, True, p, drop)
    elif ckind == "highway":
        yc = O.highway_conv1d(sd, "c", h, k, d, True, p, drop)
    elif ckind == "convT":
        yc = O.conv_transpose1d_k2s2(sd, "c", h)
    else:
        yc = O.conv1d(sd, "c", h)
        if ckind == "relu1x1":
            yc = torch.relu(yc)
    assert rel_err(y1, yc.detach()) < 5e-5
    gc = torch.autograd.grad((yc * wgt).sum(), cin)
    for n, a, b in zip(names, g1, gc):
        assert rel_err(a, b) < 8e-5, n


def test_fused_gate_backward_serves_the_kernels_it_names(dev, gemm_mode):
    """variant ids: 256 x 256 kernel 3111 (fused tail) / 3116 (+ pair-word input), 128-wide tiles 36xx; wgrad 3047 (pair-word g,
    one window for the three taps, staging between the MFMAs) / 3046 (fp32 g)"""
    from deepvoice3_pytorch_amd import ops, _lib
    h = _lib.lib()
    rng = np.random.RandomState(5)
    B, C, T, k = 16, 512, 1024, 3           # 2 x 64 = 128 tiles of 256 x 256: the 256 x 256 kernel's threshold
    x = torch.from_numpy(rng.randn(B, C, T).astype(np.float32)).to(dev).requires_grad_(True)
    layers = []
    for i in range(3):
        v, g, b = _gated_params(rng, C, k)
        layers.append([t.to(dev).requires_grad_(True) for t in (v, g, b)])
    cfg = ops.LayerCfg(k=k, dil=1, causal=False, mode=ops.EPI_GLU, residual=True)
    seen = []
    real_conv, real_wgrad = ops.conv_gemm, ops.wgrad_gemm

    def spy_conv(*a, **kw):
        y = real_conv(*a, **kw)
        if kw.get("mode") == ops.EPI_DGRAD:
            seen.append(("dgrad", h.dv3_debug_get(10), kw.get("gate") is not None, bool(kw.get("x_pair"))))
        return y

    def spy_wgrad(*a, **kw):
        out = real_wgrad(*a, **kw)
        seen.append(("wgrad", h.dv3_debug_get(11), bool(kw.get("g_pair"))))
        return out
    ops.conv_gemm, ops.wgrad_gemm = spy_conv, spy_wgrad
    try:
        t = x
        for i, (v, g, b) in enumerate(layers):
            t = ops.conv_layer(t, v, g, b, cfg)
            if i < 2:
                ops.mark_sole_consumer(t)
        t.sum().backward()
        torch.cuda.synchronize()
    finally:
        ops.conv_gemm, ops.wgrad_gemm = real_conv, real_wgrad
    dg = [s for s in seen if s[0] == "dgrad"]
    wg = [s for s in seen if s[0] == "wgrad"]
    # backward order: layer 2 (its own stand-alone gate backward wrote pair words; fused tail for layer 1), layer 1
    # (pair words from layer 2's tail; fused tail for layer 0), layer 0 (pair words from layer 1's tail)
    assert [s[1] for s in dg] == [3116, 3116, 3106], dg
    assert [s[2:] for s in dg] == [(True, True), (True, True), (False, True)], dg
    assert [s[1:] for s in wg] == [(3047, True)] * 3, wg
    # ... and with pair words off the fp32 forms
    prev, ops.pair_words = ops.pair_words, False
    try:
        del seen[:]
        ops.conv_gemm, ops.wgrad_gemm = spy_conv, spy_wgrad
        t = x
        for i, (v, g, b) in enumerate(layers):
            t = ops.conv_layer(t, v, g, b, cfg)
            if i < 2:
                ops.mark_sole_consumer(t)
        t.sum().backward()
        torch.cuda.synchronize()
    finally:
        ops.conv_gemm, ops.wgrad_gemm = real_conv, real_wgrad
        ops.pair_words = prev
    assert [s[1:] for s in seen if s[0] == "dgrad"] == [(3111, True, False), (3111, True, False), (3101, False, False)], seen
    assert [s[1:] for s in seen if s[0] == "wgrad"] == [(3046, False)] * 3, seen


def test_pair_words_are_the_operands_the_gradient_gemms_build_themselves(dev, gemm_mode):
    """stand-alone gate backward writing pair words (ops.pair_words, the default) against the fp32 pre-gate gradient:
    every gradient of the chain bit for bit"""
    from deepvoice3_pytorch_amd import ops
    rng = np.random.RandomState(21)
    for (B, C, T, k, d, kind) in ((3, 96, 150, 3, 3, "glu"), (2, 256, 804, 3, 1, "glu"), (2, 128, 200, 3, 9, "highway")):
        pv, pg, pb = _gated_params(rng, C, k)
        x = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
        cfg = ops.LayerCfg(k=k, dil=d, causal=(kind == "highway"), mode=ops.EPI_GLU if kind == "glu" else ops.EPI_HIGHWAY,
                           residual=(kind == "glu"), p=0.1, training=True, site="pw")
        out = []
        for pw in (True, False):
            prev, ops.pair_words = ops.pair_words, pw
            try:
                ops.dropout_state.manual_seed(5)
                gin = [t.clone().to(dev).requires_grad_(True) for t in (x, pv, pg, pb)]
                y = ops.conv_layer(*gin, cfg)
                out.append([t.cpu() for t in torch.autograd.grad(y.square().sum(), gin)])
            finally:
                ops.pair_words = prev
        for a, b in zip(*out):
            assert torch.equal(a, b)


def test_another_consumer_falls_back_to_the_standalone_kernel(dev, gemm_mode):
    """a producer whose output feeds two consumers: autograd sums the two gradients into a new tensor, the fused result
    of the marked consumer is dropped and dv3_gate_bwd_f32 runs -- same gradients as without any marking"""
    from deepvoice3_pytorch_amd import ops
    rng = np.random.RandomState(11)
    B, C, T, k = 2, 64, 96, 3
    pv, pg, pb = _gated_params(rng, C, k)
    cv, cg, cb = _gated_params(rng, C, k)
    x = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
    cfg = ops.LayerCfg(k=k, dil=1, causal=False, mode=ops.EPI_GLU, residual=True)

    def run(mark):
        gin = [t.clone().to(dev).requires_grad_(True) for t in (x, pv, pg, pb, cv, cg, cb)]
        hh = ops.conv_layer(gin[0], gin[1], gin[2], gin[3], cfg)
        if mark:
            ops.mark_sole_consumer(hh)                  # a promise the graph below breaks
        y = ops.conv_layer(hh, gin[4], gin[5], gin[6], cfg) + 0.5 * hh
        before = dict(ops.gate_fuse_stats)
        grads = torch.autograd.grad(y.square().sum(), gin)
        torch.cuda.synchronize()
        return [t.cpu() for t in grads], {kk: ops.gate_fuse_stats[kk] - before[kk] for kk in before}
    g1, st1 = run(True)
    g0, st0 = run(False)
    assert st1["fused"] == 0 and st1["standalone"] == 2, st1
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)
