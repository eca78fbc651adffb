# coding: utf-8
"""The 1 x 1 convolutions / Linear layers of the benchmark step (attention projections, prenet, first / last layers of the
stacks: ~30 launches, ~0.9 ms of a 15.3 ms step at 17-50 TF) under every tile configuration of the generic split-operand
tap-GEMM (tile_hint 21..29) against the picker's choice: is the picker right for K = 80..512, and where is the floor?"""
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


def timeit(fn, iters=40, settle=20):
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


B = 64
for (Cin, M, T) in [(256, 256, 201), (256, 256, 150), (256, 80, 201), (80, 256, 201), (256, 512, 201), (512, 256, 150), (256, 512, 804), (512, 513, 804)]:
    torch.manual_seed(0)
    x = torch.randn(B, Cin, T, device=dev)
    v = torch.randn(M, Cin, 1, device=dev) / math.sqrt(Cin)
    g = v.reshape(M, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.randn(M, device=dev) * 0.1
    pk = ops.pack_weights(v, g, glu_cg=0, need_bwd=False)
    y = torch.empty(B, M, T, device=dev)
    kw = dict(B=B, Cin=Cin, Tin=T, M=M, Tout=T, J=1, dil=1, padL=0, mode=ops.EPI_LINEAR, bias=bias, a_split=pk.fwd_s, y=y)
    t0 = timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw))
    v0 = L.dv3_debug_get(10)
    res = []
    for hint in range(21, 30):
        try:
            t = timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, tile_hint=hint, **kw))
            res.append("%d:%.1f" % (hint - 20, t))
        except Exception as e:
            res.append("%d:-" % (hint - 20))
    fl = 2.0 * B * T * M * Cin
    print("Cin=%3d M=%3d T=%3d  picker v%d %.1f us (%.0f TF) | %s" % (Cin, M, T, v0, t0, fl / t0 / 1e6, "  ".join(res)), flush=True)
