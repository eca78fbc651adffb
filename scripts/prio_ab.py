# coding: utf-8
"""A/B of the wave-priority knobs (dv3_debug_set 14: ping-pong tap-GEMM, 15: all-taps wgrad) at the north-star
shape; interleaved rounds in one process (developer measurement, not part of the product)."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deepvoice3_pytorch_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = {}
for rnd in range(rounds):
    for v in (0, 1, 2, 3, 4):
        L.dv3_debug_set(14, v)
        rf = bench.conv_roofline(dev, iters=60)
        res.setdefault(("conv", v), []).append(rf["us_per_launch"])
    L.dv3_debug_set(14, 0)
    for v in (0, 1, 2):
        L.dv3_debug_set(15, v)
        rf = bench.wgrad_roofline(dev, iters=40)
        res.setdefault(("wgrad", v), []).append(rf["us_per_launch"])
    L.dv3_debug_set(15, 0)
for k in sorted(res):
    print("%-6s prio=%d  us/launch: %s  min %.1f" % (k[0], k[1], " ".join("%.1f" % u for u in res[k]), min(res[k])))
