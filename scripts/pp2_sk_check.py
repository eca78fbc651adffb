# coding: utf-8
"""Stream-K form of the 256 x 256 k16 ping-pong tap-GEMM (conv_gemm_pp2.hip, round 4) against its tile-per-workgroup form
over the conv shapes of the benchmark step (B = 64, text 150, frames 800 -> T = 150 / 201 / 402 / 804): results (max
difference relative to the output's max: the two forms differ by fp32 summation order of the cut tiles only), run-to-run
bit identity of the stream-K form, and time of both.  dv3_debug_set(22, 0 | 1 | 2) = never | by the cost rule | always.

    python scripts/pp2_sk_check.py [quick]"""
import math
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepvoice3_pytorch_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
gemm = "f16x3"
ops.set_gemm_precision(gemm)
ops.streamk = "force"      # hand the workspace over with a forced tile too


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
shapes = [(512, 150, 1, False), (512, 150, 27, False), (256, 201, 1, True), (256, 201, 27, True), (256, 402, 3, False),
          (256, 804, 1, False), (512, 804, 3, False), (256, 1024, 1, False)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = shapes[:1] + shapes[2:3]
bad = 0
for (C, T, d, causal) in shapes:
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.randn(2 * C, device=dev) * 0.1
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True)
    ops.dropout_state.manual_seed(3)
    bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
    kb = ops.mask_bits_to_c8(bits, rs, B, C, T)
    gm = torch.randn(B, 2 * C, T, device=dev)
    padL = (k - 1) * d if causal else d
    ekw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
               residual=1, a_split=pk.fwd_s, tile_hint=30)
    mkw = dict(ekw, xmask=bits, xmask_rs=rs, xmask_c8=kb, drop_scale=1 / 0.95)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=d, padL=(k - 1) * d - padL, mode=ops.EPI_DGRAD,
               ymask=bits, ymask_rs=rs, drop_scale=1 / 0.95, a_split=pk.bwd_s, r=x, r_scale=0.7071, tile_hint=30)
    line = "C=%3d T=%4d d=%2d causal=%d " % (C, T, d, causal)
    for name, xin, lda, ah, kw, cout in (("eval", x, pk.lda, pk.a_half, ekw, C), ("train", x, pk.lda, pk.a_half, mkw, C),
                                         ("dgrad", gm, pk.ldb, 0, dkw, C)):
        outs, ts, vs = {}, {}, {}
        for sk in (0, 2):
            L.dv3_debug_set(22, sk)
            y = torch.empty(B, cout, T, device=dev)
            ab = torch.empty(B, 2 * C, T, device=dev) if name == "train" else None
            ops.conv_gemm(xin, None, lda, ah, y=y, ab=ab, **kw)
            vs[sk] = L.dv3_debug_get(10)
            torch.cuda.synchronize()
            outs[sk] = (y, ab)
            ts[sk] = timeit(lambda: ops.conv_gemm(xin, None, lda, ah, y=y, ab=ab, **kw))
        y2 = torch.empty(B, cout, T, device=dev)
        ops.conv_gemm(xin, None, lda, ah, y=y2, ab=(torch.empty_like(outs[2][1]) if outs[2][1] is not None else None), **kw)
        torch.cuda.synchronize()
        same = torch.equal(y2, outs[2][0])
        diff = float((outs[0][0] - outs[2][0]).abs().max() / outs[0][0].abs().max())
        dab = float((outs[0][1] - outs[2][1]).abs().max() / outs[0][1].abs().max()) if outs[0][1] is not None else 0.0
        ok = same and diff < 2e-6 and dab < 2e-6 and math.isfinite(diff)
        bad += 0 if ok else 1
        line += "| %s %d %6.1f -> %d %6.1f us (%.2f) diff %.1e%s%s " % (name, vs[0], ts[0], vs[2], ts[2], ts[2] / ts[0], max(diff, dab),
                                                                      "" if same else " NOT-REPEATABLE", "" if ok else " BAD")
    print(line, flush=True)
L.dv3_debug_set(22, 1)
print("FAILED: %d" % bad if bad else "all stream-K results within 2e-6 of the tile-per-workgroup kernel and repeatable")
