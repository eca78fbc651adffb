# coding: utf-8
"""GPU scratch check of the bf16x3 tap-GEMM: error vs an fp64 reference next to the exact fp32
kernel, and launch time at the north-star shape.  Not a test; see tests/ for the parity suite."""
import math, sys, os
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvoice3_pytorch_amd import ops

dev = torch.device("cuda:0")


def ref_glu(x, w, bias, k, d, causal, residual):
    x64, w64, b64 = x.double(), w.double(), bias.double()
    pad = (k - 1) * d if causal else (k - 1) // 2 * d
    y = F.conv1d(x64, w64, b64, padding=pad, dilation=d)[:, :, :x.shape[2]]
    a, g = y.split(y.shape[1] // 2, dim=1)
    out = a * torch.sigmoid(g)
    return (out + x64) * math.sqrt(0.5) if residual else out


def run(B, C, T, k, d, causal, hint):
    rng = np.random.RandomState(C + T + k + d)
    x = torch.from_numpy(rng.randn(B, C, T).astype(np.float32))
    v = torch.from_numpy(rng.randn(2 * C, C, k).astype(np.float32) * math.sqrt(4.0 / (k * C)))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1) * torch.from_numpy(rng.uniform(0.8, 1.2, (2 * C, 1, 1)).astype(np.float32))
    bias = torch.from_numpy(rng.uniform(-0.1, 0.1, 2 * C).astype(np.float32))
    w = g * v / v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1)
    want = ref_glu(x, w, bias, k, d, causal, True)
    ops.set_gemm_precision("bf16x3")
    pk = ops.pack_weights(v.to(dev), g.to(dev), glu_cg=C, need_bwd=False)
    xg = x.to(dev)
    padL = (k - 1) * d if causal else (k - 1) // 2 * d
    res = {}
    for name, kw in (("f32", dict(tile_hint=0 if hint == 0 else max(hint - 20, 0))), ("x3", dict(a_split=pk.fwd_s, tile_hint=hint))):
        y = ops.conv_gemm(xg, pk.fwd, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d,
                          padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias.to(dev), r=xg, residual=1, **kw)
        e = (y.cpu().double() - want).abs()
        res[name] = (float(e.max() / want.abs().max()), float(e.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()))
    print("B=%d C=%d T=%d k=%d d=%d causal=%d hint=%d  f32 max %.2e rms %.2e | x3 max %.2e rms %.2e" % (
        B, C, T, k, d, causal, hint, res["f32"][0], res["f32"][1], res["x3"][0], res["x3"][1]))
    return res


def timeit(hint, dil=1, iters=20, B=64, C=256, T=1024, k=3):
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.zeros(2 * C, device=dev)
    ops.set_gemm_precision("bf16x3")
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
    y = torch.empty(B, C, T, device=dev)
    def launch():
        ops.conv_gemm(x, pk.fwd, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=dil,
                      padL=(k - 1) // 2 * dil, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x, residual=1, y=y,
                      tile_hint=hint, a_split=pk.fwd_s if (hint == 0 or hint > 20) else None)
    for _ in range(3):
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
    print("hint=%2d dil=%2d B=%d C=%d T=%d: %8.1f us  %7.1f TFLOP/s (fp32-equivalent)" % (hint, dil, B, C, T, us, fl / us / 1e6))


def mask_check(B=3, C=96, T=150, k=3, d=9):
    """dropout keep-bits path: bf16x3 vs the exact kernel on the same mask"""
    rng = np.random.RandomState(7)
    x = torch.from_numpy(rng.randn(B, C, T).astype(np.float32)).to(dev)
    v = torch.from_numpy(rng.randn(2 * C, C, k).astype(np.float32) * 0.1).to(dev)
    g = torch.from_numpy(rng.uniform(0.5, 1.5, (2 * C, 1, 1)).astype(np.float32)).to(dev)
    bias = torch.zeros(2 * C, device=dev)
    ops.set_gemm_precision("bf16x3")
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
    ops.dropout_state.manual_seed(3)
    bits, rs = ops.dropout_bits(B * C, T, 0.3, dev)
    ys = []
    for sp in (None, pk.fwd_s):
        ys.append(ops.conv_gemm(x, pk.fwd, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d,
                                padL=(k - 1) * d, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x, residual=1, xmask=bits,
                                xmask_rs=rs, drop_scale=1 / 0.7, a_split=sp))
    print("masked x3 vs f32: max rel %.2e" % float((ys[0] - ys[1]).abs().max() / ys[0].abs().max()))


def time_wgrad(B, M, Cin, T, J=3, dil=3, split=True, masked=True, iters=10, big=0):
    from deepvoice3_pytorch_amd import _lib
    _lib.call("dv3_debug_set", 2, 2 if big else 1)
    g = torch.randn(B, M, T, device=dev)
    x = torch.randn(B, Cin, T, device=dev)
    bm = 256 if big else 128
    tiles = ((M + bm - 1) // bm) * ((Cin + 127) // 128) * J
    S = ops._ksplit_count(B * ((T + 31) // 32), tiles, slots=256 if big else 512) if split else ops._slab_count(B, tiles)
    bits, rs = (ops.dropout_bits(B * Cin, T, 0.05, dev) if masked else (None, 0))
    out = torch.empty(S, J, M, Cin, device=dev)
    def launch():
        ops.wgrad_gemm(g, x, B=B, M=M, Cin=Cin, T=T, Tin=T, J=J, dil=dil, padL=dil, n_slabs=S, xmask=bits,
                       xmask_rs=rs, drop_scale=1 / 0.95, out=out, split_bf16=split, k_split=split)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * B * T * M * Cin * J
    print("wgrad split=%d big=%d B=%d M=%d Cin=%d T=%d S=%d: %8.1f us  %7.1f TFLOP/s" % (split, big, B, M, Cin, T, S, us, fl / us / 1e6))
    _lib.call("dv3_debug_set", 2, 0)


if __name__ == "__main__":
    for rep in range(2):      # interleaved A/B: the first seconds of a process run at lower clocks
        for big in (0, 1):
            time_wgrad(64, 512, 256, 800, big=big)
            time_wgrad(64, 1024, 512, 150, big=big)
    mask_check()
    for shape in [(3, 64, 200, 3, 1, False), (3, 96, 150, 3, 27, True), (3, 20, 37, 5, 3, False), (3, 128, 513, 3, 9, True),
                  (2, 256, 1024, 3, 1, False), (2, 512, 150, 3, 27, False)]:
        for hint in (0, 21, 22, 28, 29):
            try:
                run(*shape, hint)
            except RuntimeError as e:
                print('skip', shape, hint, str(e)[-60:])
    from deepvoice3_pytorch_amd import _lib
    for abl in ():
        _lib.call("dv3_debug_set", 1, abl)
        print("ablation", abl, end=": ")
        timeit(21, 1)
    _lib.call("dv3_debug_set", 1, 0)
    for hint in (0, 21, 29, 21, 29):
        for dil in (1, 27):
            timeit(hint, dil)
    for h in ():
        timeit(h, 3, B=64, C=512, T=150)
        timeit(h, 3, B=64, C=512, T=800)
        timeit(h, 3, B=64, C=256, T=800)
    timeit(0, 3, B=16, C=256, T=800)
    timeit(0, 3, B=64, C=512, T=150)
