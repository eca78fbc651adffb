# coding: utf-8
"""Round 6: dv3_gate_bwd_f32 with 16-byte accesses for ANY T (head / quads / tail per row, dv3_debug_set(55, 1), default)
against the form of rounds 3-6 (16 bytes only for gated layers with T % 4 == 0, everything else 4 bytes per lane) at the
shapes of a deepvoice3_ljspeech / nyanko step.  Per shape: element gradients bit for bit, row sums to 1e-6 of their
scale (the order of the partial sums changes), stand-alone time of both (graph-timed, r5_common.graph_time)."""
import sys

import torch

from r5_common import dev, graph_time, L
from deepvoice3_pytorch_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shapes = [("glu", 512, 150), ("glu", 512, 152), ("glu", 256, 201), ("glu", 256, 402), ("glu", 256, 804), ("glu", 512, 804),
          ("glu", 64, 203), ("highway", 256, 201), ("highway", 512, 150), ("glu16", 256, 201), ("glu16", 512, 150),
          ("relu", 512, 804), ("sigmoid", 513, 804), ("sigmoid", 513, 801), ("relu", 256, 201), ("linear", 256, 201),
          ("relu", 128, 201), ("linear", 80, 201), ("softsign", 255, 67), ("glu", 8, 3), ("relu", 5, 2), ("glu", 4, 1)]
MODES = {"glu": ops.EPI_GLU, "glu16": ops.EPI_GLU, "highway": ops.EPI_HIGHWAY, "relu": ops.EPI_RELU,
         "sigmoid": ops.EPI_SIGMOID, "linear": ops.EPI_LINEAR, "softsign": ops.EPI_SOFTSIGN}
print("B = %d" % B)
print("%-8s %5s %5s | %8s %8s %6s | %7s %7s | %s" % ("mode", "C", "T", "us old", "us new", "ratio", "MB", "TB/s", "check"))
bad = 0
for pair in (False, True):
    for mode, C, T in shapes:
        gated = mode in ("glu", "glu16", "highway")
        if pair and not gated:
            continue
        torch.manual_seed(C * 1000 + T)
        dy = torch.randn(B, C, T, device=dev)
        ab = torch.randn(B, 2 * C if gated else C, T, device=dev)
        if mode == "sigmoid":
            ab = torch.sigmoid(ab)
        if mode == "glu16":
            ab = ab.to(torch.bfloat16)
        x = torch.randn(B, C, T, device=dev) if mode == "highway" else None

        def fn():
            return ops.gate_bwd(dy, None if mode == "linear" else ab, x, B=B, C=C, T=T, mode=MODES[mode],
                                residual=1 if mode.startswith("glu") else 0, pair=pair, want_dres=mode == "highway",
                                alpha=0.7 if not gated else 1.0)
        res, us = {}, {}
        for sw in (0, 1):
            L.dv3_debug_set(55, sw)
            res[sw] = [None if t is None else t.clone() for t in fn()]
            us[sw] = graph_time(fn)
        L.dv3_debug_set(55, 1)
        ok = True
        for k in (0, 1):
            a0, a1 = res[0][k], res[1][k]
            if a0 is not None and not torch.equal(a0.view(torch.int32), a1.view(torch.int32)):
                ok = False
        p0, p1 = res[0][2], res[1][2]
        perr = float((p0 - p1).abs().max() / p0.abs().max().clamp_min(1e-30))
        if perr > 2e-6:
            ok = False
        bad += 0 if ok else 1
        n = B * C * T * 4
        if gated:
            mb = n * ((4 if mode == "glu16" else 5) + (2 if mode == "highway" else 0)) / 1e6
        else:
            mb = n * (2 if mode == "linear" else 3) / 1e6
        print("%-8s %5d %5d | %8.1f %8.1f %6.3f | %7.1f %7.2f | %s sums %.1e%s" % (
            mode, C, T, us[0], us[1], us[1] / us[0], mb, mb / us[1], "bit-identical" if ok else "MISMATCH", perr,
            "  pair" if pair else ""))
print("mismatches: %d" % bad)
sys.exit(1 if bad else 0)
