# coding: utf-8
"""Why is the replayed deepvoice3_vctk bf16 step 10 % faster with a world-size-1 RCCL group armed than without one
(profiles/r04c_bench_line.json: 12.38 vs 13.81 ms; the other two presets: +0.2 / +0.6 %)?  Same process, same weights,
same batch: ms per step, the loss after every step, segment count; run under rocprofv3 --kernel-trace --stats for the
per-kernel call counts of each mode.

    python scripts/r4_vctk_group.py nogroup|group|both [preset gemm]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deepvoice3_pytorch_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "both"
preset = sys.argv[2] if len(sys.argv) > 2 else "deepvoice3_vctk"
gemm = sys.argv[3] if len(sys.argv) > 3 else "bf16"
modes = ["nogroup", "group"] if which == "both" else [which]
for mode in modes:
    pg = None
    if mode == "group":
        bench._world1_group()
        import torch.distributed as tdist
        pg = tdist.group.WORLD
    for graph in (True, False):
        r = bench.TrainRun(dev, pg, 0, 1, preset, gemm, 64, 150, 800, graph=graph)
        losses = []
        for _ in range(30):
            losses.append(r.step())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            scal = r.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        ls = [float(s["loss"]) for s in losses]
        gn = float(scal["grad_norm"])
        seg = (len(r.runner.segs), r.runner.chunk) if r.runner is not None and r.runner.split else None
        print("%s %s %-8s %-6s %.3f ms/step  forks %d  segments/chunk %s  loss[0,1,2,9,29] %.5f %.5f %.5f %.5f %.5f  grad_norm %.5f" % (
            preset, gemm, mode, "replay" if graph else "eager", ms, ops.SideStream.forks_last, seg, ls[0], ls[1], ls[2], ls[9], ls[29], gn), flush=True)
        r.close()
