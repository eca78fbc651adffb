# coding: utf-8
"""timing-only ablations of the one-wave-per-SIMD tap-GEMM (dv3_debug_set(13, v)) at the north-star shape"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvoice3_pytorch_amd import ops, _lib
from scripts.planes_ab import timeit, x, v, g, bias, B, C, T, k, dev, lib
ops.set_gemm_precision("f16x3")
pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True)
y = torch.empty(B, C, T, device=dev)
kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x, residual=1, a_split=pk.fwd_s, y=y, tile_hint=30)
names = {0: "full", 1: "no epilogue", 2: "no global fetches", 3: "no conversions / LDS stores", 4: "no fragment reads", 5: "MFMAs only (+ prologue, epilogue)", 6: "no barrier"}
for abl in (0, 1, 2, 3, 4, 5, 6, 0):
    lib.dv3_debug_set(13, abl)
    print("%-36s %.1f us" % (names[abl], timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw), iters=30, settle=20)), flush=True)
lib.dv3_debug_set(13, 0)
gm = torch.randn(B, 2 * C, T, device=dev)
dx = torch.empty(B, C, T, device=dev)
dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_DGRAD, a_split=pk.bwd_s, y=dx)
for hint in (0, 30):
    print("dgrad hint %d: %.1f us" % (hint, timeit(lambda: ops.conv_gemm(gm, None, pk.ldb, 0, tile_hint=hint, **dkw), iters=30, settle=20)))
