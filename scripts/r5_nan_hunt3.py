# coding: utf-8
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from deepvoice3_pytorch_amd import ops
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
pre = sys.argv[1]
def go(preset, gemm, graph, tag):
    try:
        run = bench.TrainRun(dev, None, 0, 1, preset, gemm, 64, 150, 800, graph=graph)
        m = run.measure(8, 3, settle_s=0.1)
        run.close()
        print(tag, preset, "graph" if graph else "eager", "ok %.3f" % m["ms_per_step"], flush=True)
    except Exception as e:
        print(tag, preset, "graph" if graph else "eager", "FAILED", str(e)[:80], flush=True)
if pre == "nosk":
    ops.streamk = False
for kind in pre.replace("nosk", "graph").split("+"):
    go("deepvoice3_ljspeech", "f16x3", kind == "graph", pre)
go("nyanko_ljspeech", "bf16", True, pre)
go("nyanko_ljspeech", "bf16", True, pre)
