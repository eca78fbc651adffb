# coding: utf-8
"""Round 5: whole steps with the k-split form of the 128 x 64 split tile chosen by the dispatcher's rule
(dv3_debug_set(44, 1): grids of at most 256 tiles, at least 8 k-steps) against the one-group loop only (44, 0),
replayed, alternating in one process: the preset's own batch 16 (where every layer's grid is small) and the
benchmark's batch 64 (where almost none is)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from deepvoice3_pytorch_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for preset, gemm, B in (("deepvoice3_ljspeech", "f16x3", 16), ("deepvoice3_ljspeech", "f16x3", 64), ("nyanko_ljspeech", "f16x3", 16)):
    res = {}
    for rnd in range(3):
        for v in (0, 1):
            L.dv3_debug_set(44, v)
            run = bench.TrainRun(dev, None, 0, 1, preset, gemm, B, 150, 800, graph=True)
            m = run.measure(15, 5, settle_s=0.5)
            run.close()
            res.setdefault(v, []).append(round(m["ms_per_step"], 3))
    L.dv3_debug_set(44, 1)
    print(preset, gemm, "B=%d" % B, "one group only", res[0], " k-split by the rule", res[1], flush=True)
