# coding: utf-8
"""Where the stream-K form of conv_gemm_pp2 spends its time beyond its share of the chunks: timing-only ablations
(dv3_debug_set(26, bits): 1 no hand-over stores, 2 no hand-over loads, 4 no flag wait; results are wrong with any bit set)
at the encoder shape of the benchmark step (B=64, C=512, T=150: 152 tiles x 16 chunks over 256 CUs)."""
import math
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepvoice3_pytorch_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
ops.set_gemm_precision("f16x3")
ops.streamk = "force"      # hand the workspace over with a forced tile too


def timeit(fn, iters=30, settle=20):
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


B, k = 64, 3
for (C, T, d) in [(512, 150, 1), (256, 402, 3), (512, 804, 3)]:
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.randn(2 * C, device=dev) * 0.1
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
    y = torch.empty(B, C, T, device=dev)
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=d, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
              residual=1, a_split=pk.fwd_s, tile_hint=30, y=y)
    for rnd in range(2):
        L.dv3_debug_set(22, 0)
        t0 = timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw))
        L.dv3_debug_set(22, 2)
        res = []
        for abl in (0, 1, 2, 4, 7):
            L.dv3_debug_set(26, abl)
            res.append((abl, timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw))))
        L.dv3_debug_set(26, 0)
        print("C=%d T=%d round %d: tile-per-workgroup %.1f us | stream-K " % (C, T, rnd, t0) +
              "  ".join("abl %d: %.1f" % r for r in res), flush=True)
L.dv3_debug_set(22, 1)
