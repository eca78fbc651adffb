# coding: utf-8
"""GraphedTrainer(split_streams): step time against the number of fork points per segment (`chunk`): small chunks let a
side segment start sooner behind its step-stream segment (more overlap), large chunks cost fewer graph launches."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deepvoice3_pytorch_amd import train_step  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=12, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3


for preset, gemm, B in (("deepvoice3_ljspeech", "f16x3", 64), ("nyanko_ljspeech", "bf16", 64), ("deepvoice3_ljspeech", "f16x3", 16)):
    r = bench.TrainRun(dev, None, 0, 1, preset, gemm, B, 150, 800, graph=False)
    ms, host = timed(lambda: r.step())
    print("%s %s B=%d eager: %.3f ms (host loop %.2f)" % (preset, gemm, B, ms, host), flush=True)
    for rnd in range(2):
        for chunk in (2, 4, 6, 10, 20, 1000):
            g = train_step.GraphedTrainer(r.trainer, r.batch, warmup=1, split_streams=True, chunk=chunk)
            ms, host = timed(lambda: g.step())
            print("%s %s B=%d round %d chunk %4d (%2d segments): %.3f ms (host loop %.2f)" % (preset, gemm, B, rnd, chunk, len(g.segs), ms, host), flush=True)
            g.close()
        g = train_step.GraphedTrainer(r.trainer, r.batch, warmup=1, split_streams=False)
        ms, host = timed(lambda: g.step())
        print("%s %s B=%d round %d single graph: %.3f ms (host loop %.2f)" % (preset, gemm, B, rnd, ms, host), flush=True)
        g.close()
    r.close()
