import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from deepvoice3_pytorch_amd import ops
dev = torch.device("cuda:0")
preset, gemm, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
run = bench.TrainRun(dev, None, 0, 1, preset, gemm, B, 150, 800, graph=(sys.argv[4] == "1"))
n = int(sys.argv[5]) if len(sys.argv) > 5 else 3
for i in range(n):
    s = run.step()
    if n <= 3:
        torch.cuda.synchronize()
        print("step", i, float(s["loss"]), flush=True)
torch.cuda.synchronize()
print("OK", float(s["loss"]))
