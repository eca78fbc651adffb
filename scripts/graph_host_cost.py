# coding: utf-8
"""Host time of one whole-step hipGraph replay when the previous replay has already finished (so the call does not
wait on its own in-flight executable) against back-to-back replays and the eager step.  Developer tool."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
CASES = (("deepvoice3_ljspeech", "f16x3"), ("nyanko_ljspeech", "bf16"))
if len(sys.argv) > 1:
    CASES = tuple(c for c in CASES if c[0].startswith(sys.argv[1]))
MODES = (True, False) if len(sys.argv) <= 2 else (sys.argv[2] == "graph",)
for preset, gemm in CASES:
    for graph in MODES:
        run = bench.TrainRun(dev, None, 0, 1, preset, gemm, 64, 150, 800, graph=graph)
        for _ in range(3):
            run.step()
        torch.cuda.synchronize()
        res = {}
        for mode in ("back_to_back", "synced"):
            host = 0.0
            t0 = time.perf_counter()
            for _ in range(10):
                h0 = time.perf_counter()
                run.step()
                host += time.perf_counter() - h0
                if mode == "synced":
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            res[mode] = (host / 10 * 1e3, (time.perf_counter() - t0) / 10 * 1e3)
        print("%s %s graph=%d: back-to-back host %.2f ms (step %.2f) | synced-before-launch host %.2f ms (step %.2f)" % (
            preset, gemm, graph, res["back_to_back"][0], res["back_to_back"][1], res["synced"][0], res["synced"][1]),
            flush=True)
        run.close() if hasattr(run, "close") else None
        del run
