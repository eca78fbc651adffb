# coding: utf-8
"""GPU scratch check of the ping-pong main loop of the 8-wave bf16x3 tap-GEMM tiles (dv3_debug_set 3):
bitwise comparison against the in-phase loop (same MFMA order => identical results), error vs fp64,
and interleaved launch times at the north-star shape.  Not a test; see tests/ for the parity suite."""
import math, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvoice3_pytorch_amd import ops, _lib
from x3_check import ref_glu, dev


def pp(mode):
    """0 = in-phase, 1 = ping-pong main loop of the 8-wave tiles"""
    _lib.call("dv3_debug_set", 3, int(mode))


def parity(B, C, T, k, d, causal, hint, masked, terms=3):
    rng = np.random.RandomState(C + T + k + d)
    x = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
    v = torch.from_numpy(rng.randn(2 * C, C, k).astype(np.float32) * math.sqrt(4.0 / (k * C)))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1) * torch.from_numpy(rng.uniform(0.8, 1.2, (2 * C, 1, 1)).astype(np.float32))
    bias = torch.from_numpy(rng.uniform(-0.1, 0.1, 2 * C).astype(np.float32))
    w = g * v / v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1)
    ops.set_gemm_precision("bf16x3" if terms == 3 else "bf16")
    pk = ops.pack_weights(v.to(dev), g.to(dev), glu_cg=C, need_bwd=False)
    xg = x.to(dev)
    padL = (k - 1) * d if causal else (k - 1) // 2 * d
    kw = {}
    if masked:
        ops.dropout_state.manual_seed(3)
        bits, rs = ops.dropout_bits(B * C, T, 0.3, dev)
        kw = dict(xmask=bits, xmask_rs=rs, drop_scale=1 / 0.7)
    ys = []
    for on in (0, 1):
        pp(on)
        ys.append(ops.conv_gemm(xg, pk.fwd, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d,
                                padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias.to(dev), r=xg, residual=1,
                                a_split=pk.fwd_s, tile_hint=hint, **kw))
    same = all(bool(torch.equal(ys[0], y)) for y in ys[1:])
    msg = "B=%d C=%d T=%d k=%d d=%d causal=%d hint=%d masked=%d terms=%d: pp == in-phase bitwise %s" % (
        B, C, T, k, d, causal, hint, masked, terms, same)
    if not masked:
        want = ref_glu(x, w, bias, k, d, causal, True)
        e = (ys[1].cpu().double() - want).abs()
        msg += "  | vs fp64 max %.2e" % float(e.max() / want.abs().max())
    if not same:
        msg += "  MAXDIFF %.3e" % float((ys[0] - ys[1]).abs().max())
    print(msg)
    return same


def timeit(hint, on, dil=1, iters=50, B=64, C=256, T=1024, k=3, masked=False):
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.zeros(2 * C, device=dev)
    ops.set_gemm_precision("bf16x3")
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
    y = torch.empty(B, C, T, device=dev)
    kw = {}
    if masked:
        bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
        kw = dict(xmask=bits, xmask_rs=rs, drop_scale=1 / 0.95)
    pp(on)

    def launch():
        ops.conv_gemm(x, pk.fwd, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=dil,
                      padL=(k - 1) // 2 * dil, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x, residual=1, y=y,
                      tile_hint=hint, a_split=pk.fwd_s, **kw)
    for _ in range(10):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * B * T * 2 * C * k * C
    print("hint=%2d pp=%d dil=%2d k=%d masked=%d B=%d C=%d T=%d: %8.1f us  %7.1f TFLOP/s (fp32-equivalent)" % (
        hint, on, dil, k, masked, B, C, T, us, fl / us / 1e6))
    return us


if __name__ == "__main__":
    ok = True
    for shape in [(3, 64, 200, 3, 1, False), (3, 96, 150, 3, 27, True), (3, 20, 37, 5, 3, False), (3, 128, 513, 3, 9, True),
                  (2, 256, 1024, 3, 1, False), (2, 512, 150, 3, 27, False), (4, 256, 300, 1, 1, False), (2, 24, 700, 1, 1, False)]:
        for hint in (28, 29):
            for masked in (0, 1):
                ok &= parity(*shape, hint, masked)
        ok &= parity(*shape, 29, 1, terms=1)
    print("PARITY", "OK" if ok else "FAILED")
    for rep in range(3):      # interleaved A/B: the first seconds of a process run at lower clocks
        for hint in (29, 28):
            for mode in (0, 1):
                timeit(hint, mode, 1)
    for mode in (0, 1, 0, 1):
        timeit(29, mode, 27)
    for mode in (0, 1, 0, 1):
        timeit(29, mode, 1, masked=True)
    pp(1)
