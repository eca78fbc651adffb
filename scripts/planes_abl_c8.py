# coding: utf-8
"""timing-only ablations (dv3_debug_set(6, v); results are wrong by design) of the single-term planes tap-GEMM on c8
tensors at the north-star shape: what a launch costs without its LDS stores / fragment reads / MFMAs / epilogue /
global fetches / barriers"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvoice3_pytorch_amd import ops
from scripts.planes_ab import timeit, x, v, g, bias, B, C, T, k, dev, lib

ops.set_gemm_precision("bf16")
pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
names = {0: "full", 1: "no LDS stores", 2: "no fragment reads", 3: "no MFMAs", 4: "no epilogue", 5: "no global fetches",
         6: "no barriers", 8: "MFMAs only"}
x8 = ops.to_c8(x)
for tile in (9, 1):
    lib.dv3_debug_set(4, tile)
    for train in (False, True):
        ab = ops._c8_empty(B, 2 * C, T, dev) if train else None
        keep = ops.dropout_keep_c8(B, C, T, 0.05, dev) if train else None
        kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x8, residual=1,
                  a_split=pk.fwd_s, ab=ab, x_c8=x8, out_c8=True, xmask_c8=keep, drop_scale=1 / 0.95 if train else 1.0)
        row = []
        for abl in (0, 1, 2, 3, 4, 5, 6, 8, 0):
            lib.dv3_debug_set(6, abl)
            print("  tile %d train=%d abl=%d ..." % (tile, train, abl), flush=True)
            row.append("%s %.1f" % (names[abl], timeit(lambda: ops.conv_gemm(None, None, pk.lda, pk.a_half, **kw), iters=40, settle=30)))
        lib.dv3_debug_set(6, 0)
        print("tile %d train=%d | %s" % (tile, train, " | ".join(row)), flush=True)
lib.dv3_debug_set(4, 0)
