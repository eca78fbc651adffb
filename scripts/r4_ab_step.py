# coding: utf-8
"""Same-process A/B of round 4's kernel switches on whole training steps (eager launches; boxes differ by several
per cent, so only interleaved runs on one box say anything):
    c8pp    dv3_debug_set(19, 128 | 0)   256 x 256 k32 ping-pong c8 tap-GEMM on / off           (bf16 presets)
    pf2     dv3_debug_set(20, 1 | 0)     wgrad_c8 operands fetched two steps ahead / one         (bf16 presets)
    rows    ops.slab_rows_default        K-split partial sums as [J][M][S][C] rows / [S][J][M][C] slabs
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deepvoice3_pytorch_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()


def settings(c8pp, pf2, rows):
    L.dv3_debug_set(19, 128 if c8pp else 0)
    L.dv3_debug_set(20, 1 if pf2 else 0)
    ops.slab_rows_default = bool(rows)


def run(preset, gemm, variants, rounds=3, steps=8):
    r = bench.TrainRun(dev, None, 0, 1, preset, gemm, 64, 150, 800, graph=False)
    acc = {name: [] for name, _ in variants}
    for name, st in variants:      # warm every variant's kernels / allocations
        settings(*st)
        for _ in range(3):
            r.step()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for name, st in variants:
            settings(*st)
            for _ in range(2):
                r.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                r.step()
            torch.cuda.synchronize()
            acc[name].append((time.perf_counter() - t0) / steps * 1e3)
    settings(True, True, True)
    r.close()
    base = np.median(acc[variants[0][0]])
    for name, _ in variants:
        print("%-20s %-6s %-34s %s ms  median %.3f  (%+.1f %%)" % (preset, gemm, name, " ".join("%.3f" % t for t in acc[name]),
                                                            np.median(acc[name]), (np.median(acc[name]) / base - 1) * 100), flush=True)


V_BF16 = [("all on", (1, 1, 1)), ("c8pp off", (0, 1, 1)), ("wgrad_c8 one-step fetch", (1, 0, 1)), ("slab layout of round 3", (1, 1, 0)),
          ("all off (round 3)", (0, 0, 0))]
V_F16 = [("rows of slabs", (1, 1, 1)), ("slab layout of round 3", (1, 1, 0))]
run("nyanko_ljspeech", "bf16", V_BF16)
run("deepvoice3_vctk", "bf16", V_BF16)
run("deepvoice3_ljspeech", "f16x3", V_F16)
