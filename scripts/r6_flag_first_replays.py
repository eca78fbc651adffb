# coding: utf-8
"""Round 6 (ABI 43): per-step wall time of the first replays after a capture, flag-ordered backward against the segments,
several fresh captures each (is a slow replay a warm-up effect of the large graphs, or a property of a capture?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for rep in range(int(os.environ.get("REPS", "5"))):
    for mode in ("0", "1"):
        os.environ["DV3_FLAG_SYNC"] = mode
        run = bench.TrainRun(dev, None, 0, 1, "deepvoice3_ljspeech", "f16x3", B, 150, 800, graph=True)
        ts = []
        for i in range(24):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run.step()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        to = run.runner.flag_timeouts() if hasattr(run.runner, "flag_timeouts") else 0
        print("rep %d DV3_FLAG_SYNC=%s  first 8: %s  | median of the last 12: %.2f ms, max %.2f | timeouts %d"
              % (rep, mode, " ".join("%.2f" % t for t in ts[:8]), sorted(ts[12:])[6], max(ts[12:]), to), flush=True)
        run.close()
