# coding: utf-8
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
preset, gemm = "nyanko_ljspeech", "bf16"
for rnd in range(3):
    for graph in (False, True):
        try:
            run = bench.TrainRun(dev, None, 0, 1, preset, gemm, 64, 150, 800, graph=graph)
            m = run.measure(15, 5, settle_s=0.2)
            run.close()
            print(os.environ.get("DV3_SIDE_PRIORITY"), os.environ.get("DV3_RF"), "graph" if graph else "eager", rnd, "ok %.3f" % m["ms_per_step"], flush=True)
        except Exception as e:
            print(os.environ.get("DV3_SIDE_PRIORITY"), os.environ.get("DV3_RF"), "graph" if graph else "eager", rnd, "FAILED", str(e)[:100], flush=True)
