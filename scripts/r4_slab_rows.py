# coding: utf-8
"""K-split partial sums of the weight gradient: round 3's [S][J][M][Cin] slabs against round 4's [J][M][S][Cin] rows
(ops.wgrad_gemm(rows_of_slabs=True)): time of the wgrad launch and of the weight-norm backward that reduces them, at the
north-star layer (B=64, M=512, Cin=256, T=1024, k=3, 32 K-slabs) and at an encoder layer (M=1024, Cin=512, T=150)."""
import math
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepvoice3_pytorch_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=40, settle=30):
    for _ in range(settle):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for mode in ("f16x3", "bf16"):
    ops.set_gemm_precision(mode)
    for (B, C, T, k) in ((64, 256, 1024, 3), (64, 512, 150, 3)):
        torch.manual_seed(0)
        M = 2 * C
        x = torch.randn(B, C, T, device=dev)
        gm = torch.randn(B, M, T, device=dev)
        v = torch.randn(M, C, k, device=dev) * 0.05
        g = v.reshape(M, -1).norm(dim=1).view(-1, 1, 1).clone()
        pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True)
        part = torch.randn(B, M, device=dev)
        tiles = ((M + 127) // 128) * ((C + 127) // 128)
        S = ops._ksplit_count(B * ((T + 31) // 32), tiles, slots=256)
        dv, dg, db = torch.zeros_like(v), torch.zeros_like(g), torch.zeros(M, device=dev)
        if mode == "bf16":
            x8, g8 = ops.to_c8(x), ops.to_c8(gm)
        res = {}
        for rows in (False, True):
            if mode == "bf16":
                wg = lambda: ops.wgrad_gemm_c8(g8, x8, B=B, M=M, Cin=C, T=T, J=k, dil=1, padL=1, n_slabs=S, rows_of_slabs=rows)
            else:
                wg = lambda: ops.wgrad_gemm(gm, x, B=B, M=M, Cin=C, T=T, Tin=T, J=k, dil=1, padL=1, n_slabs=S, split_bf16=True,
                                            k_split=True, rows_of_slabs=rows)
            slabs = wg()
            wn = lambda: ops.weight_norm_bwd(slabs, S, C, v, g, pk.scale, part, B, M, C, k, False, into=(dv, dg, db),
                                             rows_of_slabs=rows)
            res[rows] = (timeit(wg), timeit(wn), slabs)
        # same sums either way
        a = res[False][2].sum(0)
        b = res[True][2].sum(2)
        same = float((a - b).abs().max())
        print("%-6s B=%d M=%d Cin=%d T=%d S=%d: wgrad %.1f -> %.1f us, wn_bwd %.1f -> %.1f us (slab sums differ by %.2e)" % (
            mode, B, M, C, T, S, res[False][0], res[True][0], res[False][1], res[True][1], same), flush=True)
