# coding: utf-8
"""Replay under a world-size-1 RCCL group against the same steps without a group at the benchmark's sizes: gradient
norm and loss of every step (dropout streams seeded alike), all three presets.  Found in round 4 after
profiles/r04c_bench_line.json showed the deepvoice3_vctk replay 10 % FASTER with the group armed."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deepvoice3_pytorch_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
sel = sys.argv[1:] or ["deepvoice3_vctk:bf16", "deepvoice3_ljspeech:f16x3", "nyanko_ljspeech:bf16"]
bench._world1_group()
import torch.distributed as tdist  # noqa: E402
for item in sel:
    preset, gemm = item.split(":")
    for (mode, pg, graph) in (("nogroup eager", None, False), ("nogroup replay", None, True), ("group eager", tdist.group.WORLD, False),
                              ("group replay", tdist.group.WORLD, True)):
        ops.dropout_state.manual_seed(77)
        r = bench.TrainRun(dev, pg, 0, 1, preset, gemm, 64, 150, 800, graph=graph)
        out = []
        for _ in range(60):
            s = r.step()
            out.append((float(s["loss"]), float(s["grad_norm"])))
        seg = (len(r.runner.segs), r.runner.chunk) if r.runner is not None and r.runner.split else None
        bk = (r.runner.seg_buckets, r.runner.rest_buckets) if seg and pg is not None else None
        print("%s %s %-14s seg %s buckets %s  " % (preset, gemm, mode, seg, bk) + "  ".join("%.5f/%.4g" % o for o in out[:3] + out[-4:]), flush=True)
        r.close()
