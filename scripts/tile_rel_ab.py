# coding: utf-8
"""A/B of the split-kernel tile picker's relative cost of the 128x64 tile (dv3_debug_set(9, percent)) on the headline
train step (deepvoice3_ljspeech, f16x3)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from deepvoice3_pytorch_amd import _lib
import torch
for rel in (112, 250, 400, 112, 250):
    _lib.call("dv3_debug_set", 9, rel)
    run = bench.TrainRun(torch.device("cuda:0"), None, 0, 1, "deepvoice3_ljspeech", "f16x3", 64, 150, 800, False)
    m = run.measure(20, 8)
    run.close()
    print("rel2=%d: %.3f ms/step" % (rel, m["ms_per_step"]), flush=True)
_lib.call("dv3_debug_set", 9, 112)
