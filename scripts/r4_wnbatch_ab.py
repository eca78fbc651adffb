# coding: utf-8
"""Weight-norm backward batched (ops.WnBwdBatch: 8 layers per launch, flushed on the weight-gradient stream) against one
launch per layer, replayed steps of the three presets in one process.  Prompted by profiles/r04d_bench_line.json: the
deepvoice3_vctk replay under a world-1 RCCL group (where gradient-ready hooks switch the batching off) was 6 % FASTER
than the same replay without a group."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
for (preset, gemm) in (("deepvoice3_vctk", "bf16"), ("deepvoice3_ljspeech", "f16x3"), ("nyanko_ljspeech", "bf16")):
    res = {}
    for rnd in range(2):
        for batched in (True, False):
            r = bench.TrainRun(dev, None, 0, 1, preset, gemm, 64, 150, 800, graph=False)
            r.trainer.batch_wn_bwd = batched
            from deepvoice3_pytorch_amd import train_step
            r.runner = train_step.GraphedTrainer(r.trainer, r.batch, warmup=2)
            r.use_graph = True
            for _ in range(6):
                r.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(12):
                r.step()
            torch.cuda.synchronize()
            res.setdefault(batched, []).append((time.perf_counter() - t0) / 12 * 1e3)
            r.close()
    print("%s %s replay: batched %s ms | per layer %s ms" % (preset, gemm, " ".join("%.3f" % t for t in res[True]),
                                                           " ".join("%.3f" % t for t in res[False])), flush=True)
