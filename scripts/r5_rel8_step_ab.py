# coding: utf-8
"""Round 5: whole steps with the picker's relative cost of the 256 x 128 ping-pong tile (dv3_debug_set(42, percent);
shipped 93) lowered -- the per-launch census (profiles/r05_conv_census_dv3lj_b64.txt) has that tile ahead of the
128 x 256 one stand-alone for the encoder's input gradients and the 1 x 1 layers at T = 804; what counts is the step, where
those launches share the chip with the weight-gradient stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from deepvoice3_pytorch_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for preset, gemm, B in (("deepvoice3_ljspeech", "f16x3", 64), ("deepvoice3_ljspeech", "f16x3", 16)):
    res = {}
    for rnd in range(3):
        for v in (93, 86, 80):
            L.dv3_debug_set(42, v)
            run = bench.TrainRun(dev, None, 0, 1, preset, gemm, B, 150, 800, graph=True)
            m = run.measure(15, 5, settle_s=0.5)
            run.close()
            res.setdefault(v, []).append(round(m["ms_per_step"], 3))
    L.dv3_debug_set(42, 93)
    print(preset, gemm, "B=%d" % B, " ".join("rel8=%d %s" % (v, res[v]) for v in (93, 86, 80)), flush=True)
