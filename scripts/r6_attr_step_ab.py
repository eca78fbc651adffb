# coding: utf-8
"""Round 6: whole steps, replayed, alternating in one process, one attribute of deepvoice3_pytorch_amd.ops switched
(dotted paths reach into its classes: MaskPlan.enabled).
argv: attr off_value on_value [preset:gemm:B ...]   (values are Python literals)"""
import ast, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from deepvoice3_pytorch_amd import ops
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
attr, off, on = sys.argv[1], ast.literal_eval(sys.argv[2]), ast.literal_eval(sys.argv[3])
cases = [c.split(":") for c in sys.argv[4:]] or [("deepvoice3_ljspeech", "f16x3", "64"), ("deepvoice3_ljspeech", "f16x3", "16"),
                                                 ("nyanko_ljspeech", "bf16", "64"), ("deepvoice3_vctk", "bf16", "64")]
def _set(v):        # attr may be dotted: MaskPlan.enabled
    obj, names = ops, attr.split(".")
    for n in names[:-1]:
        obj = getattr(obj, n)
    setattr(obj, names[-1], v)


for preset, gemm, B in cases:
    res = {}
    for rnd in range(3):
        for v in (off, on):
            _set(v)
            run = bench.TrainRun(dev, None, 0, 1, preset, gemm, int(B), 150, 800, graph=True)
            m = run.measure(15, 5, settle_s=0.5)
            run.close()
            res.setdefault(repr(v), []).append(round(m["ms_per_step"], 3))
    _set(off)
    print(preset, gemm, "B=%s" % B, "ops.%s = %r:" % (attr, off), res[repr(off)], " = %r:" % (on,), res[repr(on)], flush=True)
