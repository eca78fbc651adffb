# coding: utf-8
"""Round 6: dv3_gate_bwd_f32 on c8 tensors with the sigmoid by v_exp_f32 + v_rcp_f32 (dv3_debug_set(56, 1), default)
against expf + a true division (rounds 3-6), at the shapes of the nyanko / deepvoice3_vctk bf16 steps: time of both
(graph-timed) and how far the bf16 results move (in bf16 ulps of the libm result)."""
import sys

import torch

from r5_common import dev, graph_time, L
from deepvoice3_pytorch_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shapes = [("glu", 256, 201), ("glu", 256, 804), ("glu", 512, 150), ("glu", 256, 402), ("highway", 256, 201),
          ("highway", 512, 150), ("highway", 256, 804), ("glu", 128, 201), ("relu", 512, 804), ("sigmoid", 513, 804)]
MODES = {"glu": ops.EPI_GLU, "highway": ops.EPI_HIGHWAY, "relu": ops.EPI_RELU, "sigmoid": ops.EPI_SIGMOID}
print("B = %d" % B)
print("%-8s %5s %5s | %8s %8s %6s | %7s %7s | %s" % ("mode", "C", "T", "us libm", "us fast", "ratio", "MB", "TB/s", "results"))
for mode, C, T in shapes:
    gated = mode in ("glu", "highway")
    torch.manual_seed(C + T)
    dy = ops.to_c8(torch.randn(B, C, T, device=dev))
    ab = torch.randn(B, 2 * C if gated else C, T, device=dev)
    if mode == "sigmoid":
        ab = torch.sigmoid(ab)
    ab = ops.to_c8(ab)
    x = ops.to_c8(torch.randn(B, C, T, device=dev)) if mode == "highway" else None

    def fn():
        return ops.gate_bwd_c8(dy, ab, x, B=B, C=C, T=T, mode=MODES[mode], residual=1 if mode == "glu" else 0,
                               want_dres=mode == "highway")
    res, us = {}, {}
    for sw in (0, 1):
        L.dv3_debug_set(56, sw)
        res[sw] = [None if t is None else t.clone() for t in fn()]
        us[sw] = graph_time(fn)
    L.dv3_debug_set(56, 1)
    notes = []
    for k, name in ((0, "dab"), (1, "dres")):
        a0, a1 = res[0][k], res[1][k]
        if a0 is None:
            continue
        f0, f1 = a0.float(), a1.float()
        diff = (a0.view(torch.int16).int() - a1.view(torch.int16).int()).abs()      # same sign and binade: ulps apart
        notes.append("%s: %.4f %% differ, max %d ulp" % (name, 100.0 * float((diff > 0).float().mean()), int(diff.max())))
        assert int(diff.max()) <= 1 or float((f0 - f1).abs().max()) < 1e-2 * float(f0.abs().max())
    p0, p1 = res[0][2], res[1][2]
    notes.append("sums %.1e" % float((p0 - p1).abs().max() / p0.abs().max()))
    n = B * C * T * 2
    mb = n * ((5 + (2 if mode == "highway" else 0)) if gated else 3) / 1e6
    print("%-8s %5d %5d | %8.1f %8.1f %6.3f | %7.1f %7.2f | %s" % (mode, C, T, us[0], us[1], us[1] / us[0], mb, mb / us[1],
                                                                  "; ".join(notes)))
