# coding: utf-8
"""tile_hint 0 (picker) vs 30 (256 x 256 k16 ping-pong) over the conv shapes of the three presets at B=64."""
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


def timeit(fn, iters=20, settle=15):
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
for (C, T, d, causal) in [(512, 150, 1, False), (512, 150, 27, False), (256, 200, 1, True), (256, 200, 27, True), (256, 400, 3, False),
                          (256, 800, 1, False), (512, 800, 3, False), (256, 1024, 1, False)]:
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.zeros(2 * C, device=dev)
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True)
    y = torch.empty(B, C, T, device=dev)
    ab = torch.empty(B, 2 * C, T, device=dev)
    ops.dropout_state.manual_seed(3)
    bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
    kb = ops.mask_bits_to_c8(bits, rs, B, C, T)
    gm = torch.randn(B, 2 * C, T, device=dev)
    dx = torch.empty(B, C, T, device=dev)
    padL = (k - 1) * d if causal else d
    mkw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
               residual=1, a_split=pk.fwd_s, y=y, xmask=bits, xmask_rs=rs, xmask_c8=kb, drop_scale=1 / 0.95, ab=ab)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=d, padL=(k - 1) * d - padL, mode=ops.EPI_DGRAD,
               ymask=bits, ymask_rs=rs, drop_scale=1 / 0.95, a_split=pk.bwd_s, y=dx, r=x, r_scale=0.7071)
    res = []
    for hint in (0, 30):
        tm = timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, tile_hint=hint, **mkw))
        vf = L.dv3_debug_get(10)
        td = timeit(lambda: ops.conv_gemm(gm, None, pk.ldb, 0, tile_hint=hint, **dkw))
        vd = L.dv3_debug_get(10)
        res.append((tm, vf, td, vd))
    print("C=%3d T=%4d d=%2d causal=%d  train fwd: %6.1f us (%d) vs pp2 %6.1f us  ratio %.2f | dgrad: %6.1f us (%d) vs pp2 %6.1f us  ratio %.2f" % (
        C, T, d, causal, res[0][0], res[0][1], res[1][0], res[1][0] / res[0][0], res[0][2], res[0][3], res[1][2], res[1][2] / res[0][2]))
