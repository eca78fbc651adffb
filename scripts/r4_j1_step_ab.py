# coding: utf-8
"""Same-process A/B of the tile picker's rule for 1 x 1 layers (dv3_debug_set(27, 1 | 0): the tile picker's rule for 1 x 1 layers) on whole training steps of the
headline configuration (eager launches, GPU bound), interleaved rounds."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deepvoice3_pytorch_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
for (preset, gemm, batch) in (("deepvoice3_ljspeech", "f16x3", 64), ("deepvoice3_ljspeech", "f16x3", 16)):
    r = bench.TrainRun(dev, None, 0, 1, preset, gemm, batch, 150, 800, graph=False)
    acc = {0: [], 1: []}
    for sk in (1, 0):
        L.dv3_debug_set(27, sk)
        for _ in range(3):
            r.step()
    torch.cuda.synchronize()
    for _ in range(4):
        for sk in (1, 0):
            L.dv3_debug_set(27, sk)
            for _ in range(2):
                r.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                scal = r.step()
            torch.cuda.synchronize()
            acc[sk].append((time.perf_counter() - t0) / 10 * 1e3)
    L.dv3_debug_set(27, 1)
    print("%s %s B=%d  new rule: %s median %.3f ms | off: %s median %.3f ms  (%+.2f %%)  loss %.4f" % (
        preset, gemm, batch, " ".join("%.3f" % t for t in acc[1]), np.median(acc[1]), " ".join("%.3f" % t for t in acc[0]),
        np.median(acc[0]), (np.median(acc[1]) / np.median(acc[0]) - 1) * 100, float(scal["loss"])), flush=True)
    r.close()

