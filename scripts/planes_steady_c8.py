# coding: utf-8
"""single-term planes tap-GEMM on c8 tensors, north-star shape: generic step loop vs the unrolled steady state
(dv3_debug_set(7, 1)), tiles 9 and 1"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvoice3_pytorch_amd import ops
from scripts.planes_ab import timeit, x, v, g, bias, B, C, T, k, dev, lib

ops.set_gemm_precision("bf16")
pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
x8 = ops.to_c8(x)
for d in (1, 27):
    for tile in (9, 1):
        lib.dv3_debug_set(4, tile)
        for train in (False, True):
            ab = ops._c8_empty(B, 2 * C, T, dev) if train else None
            keep = ops.dropout_keep_c8(B, C, T, 0.05, dev) if train else None
            kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=d, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x8, residual=1,
                      a_split=pk.fwd_s, ab=ab, x_c8=x8, out_c8=True, xmask_c8=keep, drop_scale=1 / 0.95 if train else 1.0)
            row, outs = [], []
            for steady in (0, 1, 0, 1):
                lib.dv3_debug_set(7, steady)
                row.append("%s %.1f" % ("steady" if steady else "generic", timeit(lambda: ops.conv_gemm(None, None, pk.lda, pk.a_half, **kw), iters=40, settle=30)))
                outs.append(ops.conv_gemm(None, None, pk.lda, pk.a_half, **kw))
            lib.dv3_debug_set(7, -1)
            print("d=%d tile %d train=%d | %s | identical %s" % (d, tile, train, " | ".join(row), torch.equal(outs[0], outs[1])), flush=True)
lib.dv3_debug_set(4, 0)
