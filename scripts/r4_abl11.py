# coding: utf-8
"""Round 4's first measurement: the LDS-DMA weight-panel form of the 256 x 256 k16 ping-pong
tap-GEMM (conv_gemm_pp2.hip, ABL 11) that round 3 committed compiled-but-unrun.  Needs the experiment build:

    make -C deepvoice3_pytorch_amd/csrc EXP=1 && DV3_LIBPATH=libdv3hip_exp.so python scripts/r4_abl11.py

Checks the result against the shipped kernel (bit-equality expected: same operands, same accumulation order) and
times both at the north-star shape."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepvoice3_pytorch_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()


def timeit(fn, iters=60, settle=60):
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


ops.set_gemm_precision("f16x3")
B, C, T, k = 64, 256, 1024, 3
torch.manual_seed(0)
x = torch.randn(B, C, T, device=dev)
v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
bias = torch.randn(2 * C, device=dev) * 0.1
pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
for dil in (1, 27):
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=dil, padL=dil, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
              residual=1, a_split=pk.fwd_s, tile_hint=30)
    outs = {}
    for abl in (0, 11):
        L.dv3_debug_set(13, abl)
        y = torch.empty(B, C, T, device=dev)
        ops.conv_gemm(x, None, pk.lda, pk.a_half, y=y, **kw)
        torch.cuda.synchronize()
        outs[abl] = y
    L.dv3_debug_set(13, 0)
    same = torch.equal(outs[0], outs[11])
    print("dil %2d: LDS-DMA panels vs shipped: %s (max diff %.3e)" % (
        dil, "BIT-EQUAL" if same else "DIFFERS", float((outs[0] - outs[11]).abs().max())))
    y = torch.empty(B, C, T, device=dev)
    for rnd in range(3):
        for abl in (0, 11):
            L.dv3_debug_set(13, abl)
            t = timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, y=y, **kw))
            print("dil %2d round %d  %s: %.1f us" % (dil, rnd, "LDS-DMA panels (ABL 11)" if abl else "shipped (register path)", t))
    L.dv3_debug_set(13, 0)
