# coding: utf-8
"""A/B of the planes tile picker's mid-size threshold (dv3_debug_set(8, v)) on the bf16 c8 train steps"""
import json, subprocess, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from deepvoice3_pytorch_amd import _lib
import torch
for preset in ("nyanko_ljspeech", "deepvoice3_vctk"):
    for thr in (4, 2, 1, 4, 2):
        _lib.call("dv3_debug_set", 8, thr)
        run = bench.TrainRun(torch.device("cuda:0"), None, 0, 1, preset, "bf16", 64, 150, 800, False)
        m = run.measure(20, 8)
        run.close()
        print("%s thr=%d: %.3f ms/step" % (preset, thr, m["ms_per_step"]), flush=True)
_lib.call("dv3_debug_set", 8, 4)
