# coding: utf-8
"""Same-process A/B of the stream-K form of the 256 x 256 tap-GEMM (dv3_debug_set(22, 1 | 0)) on whole training steps of the
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
        L.dv3_debug_set(22, sk)
        for _ in range(3):
            r.step()
    torch.cuda.synchronize()
    for _ in range(4):
        for sk in (1, 0):
            L.dv3_debug_set(22, sk)
            for _ in range(2):
                r.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                scal = r.step()
            torch.cuda.synchronize()
            acc[sk].append((time.perf_counter() - t0) / 10 * 1e3)
    L.dv3_debug_set(22, 1)
    print("%s %s B=%d  stream-K on: %s median %.3f ms | off: %s median %.3f ms  (%+.2f %%)  loss %.4f" % (
        preset, gemm, batch, " ".join("%.3f" % t for t in acc[1]), np.median(acc[1]), " ".join("%.3f" % t for t in acc[0]),
        np.median(acc[0]), (np.median(acc[1]) / np.median(acc[0]) - 1) * 100, float(scal["loss"])), flush=True)
    r.close()

# forward only (training mode, no autograd): where the stream-K launches are not overlapped by a second stream
from deepvoice3_pytorch_amd import ops  # noqa: E402
r = bench.TrainRun(dev, None, 0, 1, "deepvoice3_ljspeech", "f16x3", 64, 150, 800, graph=False)
t, b = r.trainer, r.batch
t.model.train()


def fwd():
    ops.prepacked = t._prepack_all()
    try:
        with torch.no_grad():
            return t.model(b.text, b.mel, speaker_ids=b.speaker_ids, text_positions=b.text_positions,
                           frame_positions=b.frame_positions, input_lengths=b.input_lengths)
    finally:
        ops.prepacked = None


acc = {0: [], 1: []}
for _ in range(4):
    for sk in (1, 0):
        L.dv3_debug_set(22, sk)
        for _ in range(3):
            fwd()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fwd()
        e1.record()
        torch.cuda.synchronize()
        acc[sk].append(e0.elapsed_time(e1) / 10)
L.dv3_debug_set(22, 1)
print("forward only B=64  stream-K on: %s median %.3f ms | off: %s median %.3f ms  (%+.2f %%)" % (
    " ".join("%.3f" % x for x in acc[1]), np.median(acc[1]), " ".join("%.3f" % x for x in acc[0]), np.median(acc[0]),
    (np.median(acc[1]) / np.median(acc[0]) - 1) * 100), flush=True)
r.close()
