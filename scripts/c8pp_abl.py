# coding: utf-8
"""Timing-only ablations of the 256 x 256 k32 ping-pong c8 tap-GEMM (csrc/conv_c8pp.hip; results are wrong by
construction).  Needs the experiment build:
    make -C deepvoice3_pytorch_amd/csrc EXP=1 && DV3_LIBPATH=libdv3hip_exp.so python scripts/c8pp_abl.py"""
import math
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepvoice3_pytorch_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
ops.set_gemm_precision("bf16")
ops.bf16_storage = True


def timeit(fn, iters=40, settle=40):
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
for (C, T) in ((256, 1024), (512, 800)):
    torch.manual_seed(0)
    x8 = ops.to_c8(torch.randn(B, C, T, device=dev))
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.zeros(2 * C, device=dev)
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False, split_only=True)
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x8,
              residual=1, a_split=pk.fwd_s, x_c8=x8, out_c8=True)
    for rnd in range(2):
        for abl, name in ((0, "full"), (1, "no MFMAs"), (2, "no staging (global fetches + LDS stores)"), (3, "no tail"),
                          (5, "no fragment reads"), (6, "no barriers")):
            L.dv3_debug_set(21, abl)
            t = timeit(lambda: ops.conv_gemm(None, None, pk.lda, pk.a_half, **kw))
            print("C=%d T=%d round %d  %-44s %7.1f us" % (C, T, rnd, name, t), flush=True)
    L.dv3_debug_set(21, 0)
