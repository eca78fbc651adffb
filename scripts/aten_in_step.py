# coding: utf-8
"""Which torch (ATen) operators still launch device work inside one training step, and from where?
VERDICT r3 next #9: "zero at::native kernels in the step".  One eager step per preset under torch.profiler with Python
stacks; prints every aten op that owns a device kernel / memcpy / memset, grouped by the innermost frame inside this
repository (forward ops) or by the autograd node that issued it (backward)."""
import collections
import os
import sys

import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
presets = sys.argv[1:] or ["deepvoice3_ljspeech:f16x3", "nyanko_ljspeech:bf16", "deepvoice3_vctk:bf16"]
for item in presets:
    preset, gemm = item.split(":")
    r = bench.TrainRun(dev, None, 0, 1, preset, gemm, 16, 60, 200, graph=False)
    for _ in range(3):
        r.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        r.step()
        torch.cuda.synchronize()
    r.close()
    ev = prof.events()
    groups = collections.Counter()
    n_dev = 0
    for e in ev:
        if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
            continue
        name = e.name
        if not (name.startswith("aten::") or "Memcpy" in name or "Memset" in name or "memcpy" in name.lower()):
            continue
        if any(c.kernels and c is not e for c in (e.cpu_children or [])):      # count the innermost op that owns the kernel
            continue
        n_dev += len(e.kernels)
        where = "?"
        for fr in (e.stack or []):
            if ROOT in fr and "scripts/" not in fr and "torch/" not in fr:
                where = fr.replace(ROOT + "/", "")
                break
        if where == "?":
            p = e.cpu_parent
            while p is not None and where == "?":
                if "Backward" in p.name or "autograd" in p.name:
                    where = "autograd: " + p.name
                p = p.cpu_parent
        groups[(name, where[:150], tuple(k.name[:60] for k in e.kernels))] += 1
    print("==== %s %s: %d device launches owned by torch operators in one step" % (preset, gemm, n_dev))
    for (name, where, kern), cnt in sorted(groups.items(), key=lambda kv: -kv[1]):
        print("%3d x %-22s %-110s %s" % (cnt, name, where, kern[0] if kern else ""))
    sys.stdout.flush()
