# coding: utf-8
"""The second backward stream on a hardware queue of its own (ops.concurrent_stream, the default) against one that shares
the step stream's queue (DV3_SIDE_STREAM_SAME_QUEUE=1: the candidate the spin-kernel probe REJECTS): replayed and eager
steps of deepvoice3_vctk bf16 and deepvoice3_ljspeech f16x3 in one process."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")


def run(preset, gemm, graph, shared):
    os.environ["DV3_SIDE_STREAM_SAME_QUEUE"] = "1" if shared else "0"
    r = bench.TrainRun(dev, None, 0, 1, preset, gemm, 64, 150, 800, graph=graph)
    for _ in range(6):
        r.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(16):
        r.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 16 * 1e3
    r.close()
    return ms


for (preset, gemm) in (("deepvoice3_vctk", "bf16"), ("deepvoice3_ljspeech", "f16x3")):
    for graph in (True, False):
        res = {False: [], True: []}
        for rnd in range(2):
            for shared in (False, True):
                res[shared].append(run(preset, gemm, graph, shared))
        print("%s %s %-6s own queue: %s ms | shared queue: %s ms" % (preset, gemm, "replay" if graph else "eager",
              " ".join("%.3f" % t for t in res[False]), " ".join("%.3f" % t for t in res[True])), flush=True)
os.environ["DV3_SIDE_STREAM_SAME_QUEUE"] = "0"
