# coding: utf-8
"""Round 5: whole steps of the bf16 presets with conv_c8pp's form chosen by the dispatcher's rule (dv3_debug_set(34, 2))
against the 8-wave form only (34, 0), replay and eager, alternating in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from deepvoice3_pytorch_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for preset, gemm in (("nyanko_ljspeech", "bf16"), ("deepvoice3_vctk", "bf16")):
    res = {}
    for rnd in range(3):
        for v in (0, 2):
            L.dv3_debug_set(34, v)
            run = bench.TrainRun(dev, None, 0, 1, preset, gemm, 64, 150, 800, graph=True)
            m = run.measure(15, 5, settle_s=0.5)
            run.close()
            res.setdefault(v, []).append(round(m["ms_per_step"], 3))
    L.dv3_debug_set(34, 2)
    print(preset, gemm, "8-wave only", res[0], " by the rule", res[2], flush=True)
