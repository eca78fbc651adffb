# coding: utf-8
"""Round 6, final tree: a soak of the configurations bench.py reports -- REPS fresh trainers per configuration, 150
replayed steps each: every loss and gradient norm finite and the loss falling; the repetitions draw their own dropout
streams (the library's seed counter moves on with every trainer of a process), so they end on different bits: the spread of
the final loss is printed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
cases = [("deepvoice3_ljspeech", "f16x3", 64), ("deepvoice3_ljspeech", "f16x3", 16), ("nyanko_ljspeech", "bf16", 64),
         ("deepvoice3_vctk", "bf16", 64), ("nyanko_ljspeech", "f16x3", 64)]
reps, steps = int(os.environ.get("REPS", "3")), int(os.environ.get("STEPS", "150"))
bad_total = 0
for preset, gemm, B in cases:
    ends, bad = [], 0
    for rep in range(reps):
        run = bench.TrainRun(dev, None, 0, 1, preset, gemm, B, 150, 800, graph=True)
        first = last = None
        for i in range(steps):
            s = run.step()
            if i % 10 == 9 or i == steps - 1:
                l, g = float(s["loss"]), float(s["grad_norm"])
                if not (l == l and g == g and abs(g) < 1e30 and abs(l) < 1e30):
                    bad += 1
                first = l if first is None else first
                last = (l, g)
        ends.append((first,) + last)
        run.close()
    finals = [e[1] for e in ends]
    bad_total += bad
    print("%-22s %-6s B=%2d  %d x %d replayed steps: non-finite checks %d; loss %.5f (step 10) -> %.5f, |g| %.4f; final loss over the repetitions %.5f .. %.5f"
          % (preset, gemm, B, reps, steps, bad, ends[0][0], ends[0][1], ends[0][2], min(finals), max(finals)), flush=True)
sys.exit(1 if bad_total else 0)
