# coding: utf-8
"""Which torch (aten) operators still launch device work inside one train step, and from which line of this package;
plus the libdv3hip entry points a step calls.  Developer tool for the launch-program work."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deepvoice3_pytorch_amd import _lib  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "deepvoice3_ljspeech"
gemm = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)
run = bench.TrainRun(dev, None, 0, 1, preset, gemm, 16, 150, 800, graph=False)
for _ in range(3):
    run.step()
torch.cuda.synchronize()

SKIP = ("aten.view", "aten.detach", "aten.t.", "aten.transpose", "aten.expand", "aten.unsqueeze", "aten.squeeze",
        "aten.select", "aten.slice", "aten.reshape", "aten._unsafe_view", "aten.alias", "aten.as_strided",
        "aten.permute", "aten.unbind", "aten.split", "aten.narrow", "aten.size", "aten.stride", "aten.is_",
        "aten.sym_", "aten._local_scalar", "aten.lift_fresh", "aten.empty", "aten.new_empty", "aten.unflatten",
        "aten._reshape_alias", "aten.chunk", "aten.view_as", "aten.empty_like", "aten.empty_strided")
sites = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if ("deepvoice3_pytorch_amd" in fr.filename or fr.filename.endswith("bench.py")):
                    where = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                    break
            shp = [tuple(a.shape) for a in args if torch.is_tensor(a)][:2]
            sites[(name, where, str(shp))] += 1
        return func(*args, **(kwargs or {}))


calls = collections.Counter()
orig = _lib.call


def call(name, *a):
    calls[name] += 1
    return orig(name, *a)


_lib.call = call
from deepvoice3_pytorch_amd import ops  # noqa: E402
if hasattr(ops, "_lib") and ops._lib is _lib:
    pass
with Log():
    run.step()
torch.cuda.synchronize()
_lib.call = orig
print("== aten ops in one step (%s %s) ==" % (preset, gemm))
for (name, where, shp), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print("%4d  %-34s %-46s %s" % (n, name, where, shp))
print("== libdv3hip calls in one step: %d ==" % sum(calls.values()))
for name, n in calls.most_common():
    print("%4d  %s" % (n, name))
