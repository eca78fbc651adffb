# coding: utf-8
"""Round 6: whole steps, replayed, alternating in one process, one ENVIRONMENT switch read at Trainer / GraphedTrainer
construction.  argv: VAR off_value on_value [preset:gemm:B ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
var, off, on = sys.argv[1], sys.argv[2], sys.argv[3]
cases = [c.split(":") for c in sys.argv[4:]] or [("deepvoice3_ljspeech", "f16x3", "64"), ("deepvoice3_ljspeech", "f16x3", "16"),
                                                 ("nyanko_ljspeech", "bf16", "64"), ("deepvoice3_vctk", "bf16", "64")]
for preset, gemm, B in cases:
    res = {}
    for rnd in range(3):
        for v in (off, on):
            os.environ[var] = v
            run = bench.TrainRun(dev, None, 0, 1, preset, gemm, int(B), 150, 800, graph=True)
            m = run.measure(15, 5, settle_s=0.5)
            nseg = len(run.runner.segs)
            run.close()
            res.setdefault(v, []).append(round(m["ms_per_step"], 3))
    print(preset, gemm, "B=%s" % B, "%s=%s:" % (var, off), res[off], " %s=%s:" % (var, on), res[on], "(%d segments)" % nseg, flush=True)
