# coding: utf-8
"""Every tile configuration of the split-operand tap-GEMMs (tile_hint 21..29 = the generic kernel's tiles, 30 = the
256 x 256 k16 ping-pong kernel) on the three-tap Conv1dGLU shapes of the benchmark step (B = 64): masked training forward
with the pre-gate save, and the input gradient, against the picker's choice (tile_hint 0)."""
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


def timeit(fn, iters=25, settle=12):
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
for (C, T, d, causal) in [(512, 150, 3, False), (256, 201, 1, True), (256, 201, 9, True), (256, 402, 3, False), (256, 804, 1, False), (512, 804, 3, False)]:
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.randn(2 * C, device=dev) * 0.1
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True)
    ops.dropout_state.manual_seed(3)
    bits, rs, kb = ops.dropout_bits_keep(B, C, T, 0.05, dev)
    gm = torch.randn(B, 2 * C, T, device=dev)
    padL = (k - 1) * d if causal else d
    y, ab, dx = torch.empty(B, C, T, device=dev), torch.empty(B, 2 * C, T, device=dev), torch.empty(B, C, T, device=dev)
    mkw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x, residual=1,
               a_split=pk.fwd_s, xmask=bits, xmask_rs=rs, xmask_c8=kb, drop_scale=1 / 0.95, y=y, ab=ab)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=d, padL=(k - 1) * d - padL, mode=ops.EPI_DGRAD, ymask=bits,
               ymask_rs=rs, drop_scale=1 / 0.95, a_split=pk.bwd_s, r=x, r_scale=0.7071, y=dx)
    for name, xin, lda, ah, kw in (("train fwd", x, pk.lda, pk.a_half, mkw), ("dgrad    ", gm, pk.ldb, 0, dkw)):
        t0 = timeit(lambda: ops.conv_gemm(xin, None, lda, ah, **kw))
        v0 = L.dv3_debug_get(10)
        res = []
        for hint in (21, 22, 28, 29, 30):
            try:
                t = timeit(lambda: ops.conv_gemm(xin, None, lda, ah, tile_hint=hint, **kw))
                res.append("%d:%.1f" % (hint - 20, t))
            except Exception:
                res.append("%d:-" % (hint - 20))
        print("C=%3d T=%3d d=%d %s picker v%d %6.1f us | %s" % (C, T, d, name, v0, t0, "  ".join(res)), flush=True)
