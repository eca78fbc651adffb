# coding: utf-8
"""timing-only ablations of the planes tap-GEMM at the north-star shape (dv3_debug_set(6, v)): what a launch costs
without its LDS stores / fragment reads / MFMAs / epilogue / global fetches / barriers (results are wrong by design)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvoice3_pytorch_amd import ops
from scripts.planes_ab import timeit, x, v, g, bias, B, C, T, k, dev, lib

ops.set_gemm_precision("f16x3")
pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
names = {0: "full", 1: "no LDS stores", 2: "no fragment reads", 3: "no MFMAs", 4: "no epilogue", 5: "no global fetches", 6: "no barriers", 7: "reads interleaved", 8: "MFMAs only", 9: "acc in AGPRs"}
for train in (False, True):
    y = torch.empty(B, C, T, device=dev)
    ab = torch.empty(B, 2 * C, T, device=dev) if train else None
    xp = ops.split_planes(x, f16=True)
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x, residual=1,
              a_split=pk.fwd_s, y=y, ab=ab, x_planes=xp)
    for tile in (9, 1):
        lib.dv3_debug_set(4, tile)
        row = []
        lib.dv3_debug_set(7, 0)
        row.append("generic loop %.1f" % timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw), iters=40, settle=40))
        lib.dv3_debug_set(7, 1)
        for abl in (0, 0, 8, 0):
            lib.dv3_debug_set(6, abl)
            row.append("%s %.1f" % (names[abl], timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw), iters=40, settle=40)))
        lib.dv3_debug_set(6, 0)
        print("train=%d tile %d | %s" % (train, tile, " | ".join(row)))
lib.dv3_debug_set(4, 0)
