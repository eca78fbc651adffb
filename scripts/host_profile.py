# coding: utf-8
"""cProfile of the eager train step's host side (where do the ~30 us per launch go?) -- developer tool."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)      # backward in this thread, so that cProfile sees it
run = bench.TrainRun(dev, None, 0, 1, "deepvoice3_ljspeech", "f16x3", 64, 150, 800, graph=False)
for _ in range(5):
    run.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    run.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(30)
