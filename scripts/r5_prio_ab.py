# coding: utf-8
"""Round 5: the weight-gradient stream at normal vs the device's least stream priority (DV3_SIDE_PRIORITY), whole steps,
eager and replay, three presets, one process per setting (the priority is read when the Trainer is built)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for preset, gemm in (("deepvoice3_ljspeech", "f16x3"), ("nyanko_ljspeech", "bf16"), ("deepvoice3_vctk", "bf16")):
    res = {}
    for rnd in range(2):
        for pr in ("normal", "low"):
            os.environ["DV3_SIDE_PRIORITY"] = pr
            for graph in (False, True):
                run = bench.TrainRun(dev, None, 0, 1, preset, gemm, 64, 150, 800, graph=graph)
                m = run.measure(15, 5, settle_s=1.0)
                run.close()
                res.setdefault((pr, graph), []).append(m["ms_per_step"])
    print(preset, gemm, {("%s %s" % (k[0], "replay" if k[1] else "eager")): [round(x, 3) for x in v] for k, v in res.items()}, flush=True)
from deepvoice3_pytorch_amd import ops
print([(r["role"], r["priority"], r["found"], r["candidates"][-1:]) for r in ops.stream_probe_log][-4:])
