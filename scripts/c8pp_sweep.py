# coding: utf-8
"""The 256 x 256 k32 ping-pong c8 tap-GEMM (csrc/conv_c8pp.hip) against the 128-row planes kernel over the conv shapes
of the three presets at B = 64 (bf16 GEMM mode, c8 storage): eval forward, masked training forward with the pre-gate
save, input gradient.  dv3_debug_set(19, 0) = planes kernel only, (19, 1) = the new kernel wherever it is eligible.
Sets the dispatch threshold (g_c8pp_min_tiles)."""
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


def timeit(fn, iters=30, settle=25):
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
shapes = [(256, 1024, 1, False, 3), (256, 1024, 27, False, 3), (512, 150, 1, False, 3), (512, 150, 27, False, 3),
          (256, 200, 1, True, 3), (256, 200, 27, True, 3), (256, 400, 3, False, 3), (256, 800, 1, False, 3),
          (512, 800, 3, False, 3), (512, 150, 1, False, 1), (256, 800, 1, False, 1), (256, 200, 1, False, 1)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = shapes[:1]
for (C, T, d, causal, k) in shapes:
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.zeros(2 * C, device=dev)
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True, split_only=True)
    x8 = ops.to_c8(x)
    ops.dropout_state.manual_seed(3)
    keep8 = ops.dropout_keep_c8(B, C, T, 0.05, dev)
    ab = ops._c8_empty(B, 2 * C, T, dev)
    gm8 = ops.to_c8(torch.randn(B, 2 * C, T, device=dev))
    padL = (k - 1) * d if causal else (k - 1) // 2 * d
    ekw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x8,
               residual=1, a_split=pk.fwd_s, x_c8=x8, out_c8=True)
    mkw = dict(ekw, xmask_c8=keep8, drop_scale=1 / 0.95, ab=ab)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=d, padL=(k - 1) * d - padL, mode=ops.EPI_DGRAD, r=x8, r_scale=0.7071,
               drop_scale=1 / 0.95, a_split=pk.bwd_s, x_c8=gm8, out_c8=True, ymask_c8=keep8)
    res = []
    for thr in (0, 1):
        L.dv3_debug_set(19, thr)
        te = timeit(lambda: ops.conv_gemm(None, None, pk.lda, pk.a_half, **ekw))
        tm = timeit(lambda: ops.conv_gemm(None, None, pk.lda, pk.a_half, **mkw))
        vf = L.dv3_debug_get(10)
        td = timeit(lambda: ops.conv_gemm(None, None, pk.ldb, 0, **dkw))
        vd = L.dv3_debug_get(10)
        res.append((te, tm, vf, td, vd))
    L.dv3_debug_set(19, 128)
    fl = 2.0 * B * T * (2 * C) * (k * C)
    tiles = ((C + 127) // 128) * ((B * T + 255) // 256)
    print("C=%3d T=%4d k=%d d=%2d causal=%d (%4d fwd tiles)  eval %6.1f -> %6.1f us (%.2f, %.0f TF)  train fwd %6.1f (%d) -> %6.1f (%d) us (%.2f) | dgrad %6.1f (%d) -> %6.1f (%d) us (%.2f)" % (
        C, T, k, d, causal, tiles, res[0][0], res[1][0], res[1][0] / res[0][0], fl / res[1][0] / 1e6, res[0][1], res[0][2], res[1][1], res[1][2],
        res[1][1] / res[0][1], res[0][3], res[0][4], res[1][3], res[1][4], res[1][3] / res[0][3]), flush=True)
