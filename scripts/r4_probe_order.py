# coding: utf-8
"""Does an eager probe before the capture slow the replay down?  deepvoice3_vctk bf16: TrainRun(graph="auto") with and
without torch.cuda.empty_cache() between the probe and the capture, against a direct capture (graph=True)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")


def run(graph, empty):
    os.environ["DV3_BENCH_EMPTY_CACHE"] = "1" if empty else "0"
    r = bench.TrainRun(dev, None, 0, 1, "deepvoice3_vctk", "bf16", 64, 150, 800, graph=graph)
    for _ in range(6):
        r.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(16):
        r.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 16 * 1e3
    used = r.use_graph
    r.close()
    return ms, used


for rnd in range(2):
    for (name, graph, empty) in (("direct capture", True, False), ("probe, then capture", "auto", False),
                                 ("probe, empty_cache, capture", "auto", True)):
        ms, used = run(graph, empty)
        print("round %d  %-28s %.3f ms/step (replay kept: %s)" % (rnd, name, ms, used), flush=True)
