# coding: utf-8
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
seq = [("deepvoice3_ljspeech", "f16x3")] * 1 + [("nyanko_ljspeech", "bf16")] * 2 + [("deepvoice3_vctk", "bf16")]
for preset, gemm in seq:
    for rnd in range(2):
        for pr in ("normal", "low"):
            os.environ["DV3_SIDE_PRIORITY"] = pr
            for graph in (False, True):
                try:
                    run = bench.TrainRun(dev, None, 0, 1, preset, gemm, 64, 150, 800, graph=graph)
                    m = run.measure(15, 5, settle_s=0.2)
                    run.close()
                    print(preset, pr, "graph" if graph else "eager", rnd, "ok %.3f" % m["ms_per_step"], flush=True)
                except Exception as e:
                    print(preset, pr, "graph" if graph else "eager", rnd, "FAILED", str(e)[:80], flush=True)
