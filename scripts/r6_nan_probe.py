import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from deepvoice3_pytorch_amd import _lib, ops
L = _lib.lib()
dev = torch.device("cuda:0")
cases = [("deepvoice3_ljspeech", "f16x3", 64), ("deepvoice3_ljspeech", "f16x3", 16), ("nyanko_ljspeech", "f16x3", 64)]
reps = int(os.environ.get("REPS", "8"))
for preset, gemm, B in cases:
    bad_runs = []
    for rep in range(reps):
        L.dv3_debug_set(47, rep & 1)
        run = bench.TrainRun(dev, None, 0, 1, preset, gemm, B, 150, 800, graph=True)
        first_bad = None
        for i in range(70):
            s = run.step()
            if i % 10 == 9:
                l, g = float(s["loss"]), float(s["grad_norm"])
                if first_bad is None and not (l == l and g == g and abs(g) < 1e30):
                    first_bad = (i, l, g)
                    bad = [n for n, p in run.model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
                    print("   ", preset, B, "rep", rep, "non-finite by step", i, (l, g), "grads:", bad[:5], "...", len(bad), flush=True)
        if first_bad:
            bad_runs.append((rep, first_bad[0]))
        run.close()
    print(preset, gemm, B, "fuse", ops.fuse_gate_bwd, "pair", ops.pair_words, "bad runs:", bad_runs, "of", reps, flush=True)
