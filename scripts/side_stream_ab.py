# coding: utf-8
"""train step with / without the weight-gradient side stream (DV3_WGRAD_STREAM), graph and eager: same process A/B"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
dev = torch.device("cuda:0")
for preset, gemm in (("deepvoice3_ljspeech", "f16x3"), ("nyanko_ljspeech", "bf16")):
    for side in ("1", "0"):
        os.environ["DV3_WGRAD_STREAM"] = side
        for graph in (True, False):
            run = bench.TrainRun(dev, None, 0, 1, preset, gemm, 64, 150, 800, graph=graph)
            m = run.measure(20, 8, settle_s=0.5)
            print("%s %s side_stream=%s graph=%d: %.3f ms/step  (host %.2f ms)  loss %.5f" % (
                preset, gemm, side, int(run.use_graph), m["ms_per_step"], m["host_enqueue_ms_per_step"], m["final_loss"]), flush=True)
            run.close()
