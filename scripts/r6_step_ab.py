# coding: utf-8
"""Round 6: whole steps, replayed, alternating in one process, one library switch at a time (dv3_debug_set(what, v)).
argv: what  off_value  on_value  [preset:gemm:B ...]   default cases: dv3lj f16x3 B=64 / B=16, nyanko f16x3 B=64"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from deepvoice3_pytorch_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
what, off, on = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cases = [c.split(":") for c in sys.argv[4:]] or [("deepvoice3_ljspeech", "f16x3", "64"), ("deepvoice3_ljspeech", "f16x3", "16"),
                                                 ("nyanko_ljspeech", "f16x3", "64")]
for preset, gemm, B in cases:
    res = {}
    for rnd in range(3):
        for v in (off, on):
            L.dv3_debug_set(what, v)
            run = bench.TrainRun(dev, None, 0, 1, preset, gemm, int(B), 150, 800, graph=True)
            m = run.measure(15, 5, settle_s=0.5)
            run.close()
            res.setdefault(v, []).append(round(m["ms_per_step"], 3))
    L.dv3_debug_set(what, on)
    print(preset, gemm, "B=%s" % B, "debug_set(%d, %d):" % (what, off), res[off], " (%d, %d):" % (what, on), res[on], flush=True)
