# coding: utf-8
"""Same-process A/B of the fused speaker-bias block path (ops.fused_speaker_bias) on whole training steps of the
multi-speaker preset (deepvoice3_vctk, bf16 and f16x3), eager launches interleaved, then each form replayed."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deepvoice3_pytorch_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for gemm in (sys.argv[1:] or ["bf16", "f16x3"]):
    r = bench.TrainRun(dev, None, 0, 1, "deepvoice3_vctk", gemm, 64, 150, 800, graph=False)
    acc = {False: [], True: []}
    for f in (True, False):
        ops.fused_speaker_bias = f
        for _ in range(3):
            r.step()
    torch.cuda.synchronize()
    for _ in range(3):
        for f in (True, False):
            ops.fused_speaker_bias = f
            for _ in range(2):
                r.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                scal = r.step()
            torch.cuda.synchronize()
            acc[f].append((time.perf_counter() - t0) / 8 * 1e3)
    r.close()
    rep = {}
    for f in (True, False):
        ops.fused_speaker_bias = f
        rr = bench.TrainRun(dev, None, 0, 1, "deepvoice3_vctk", gemm, 64, 150, 800, graph=True)
        for _ in range(5):
            rr.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(12):
            scal = rr.step()
        torch.cuda.synchronize()
        rep[f] = ((time.perf_counter() - t0) / 12 * 1e3, float(scal["loss"]))
        rr.close()
    ops.fused_speaker_bias = True
    print("deepvoice3_vctk %s B=64 eager  fused: %s median %.3f ms | per layer: %s median %.3f ms  (%+.1f %%)" % (
        gemm, " ".join("%.3f" % t for t in acc[True]), np.median(acc[True]), " ".join("%.3f" % t for t in acc[False]),
        np.median(acc[False]), (np.median(acc[True]) / np.median(acc[False]) - 1) * 100), flush=True)
    print("deepvoice3_vctk %s B=64 replay fused: %.3f ms (loss %.4f) | per layer: %.3f ms (loss %.4f)  (%+.1f %%)" % (
        gemm, rep[True][0], rep[True][1], rep[False][0], rep[False][1], (rep[True][0] / rep[False][0] - 1) * 100), flush=True)
