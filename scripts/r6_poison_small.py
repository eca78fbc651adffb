# coding: utf-8
"""Round 6: scripts/poison_check.py at SMALL batches of the preset models (B = 4, where every gated layer takes the fused
gate tail and the k-split rules pick other slab counts than at B = 64): one training forward + backward on clean memory and
again with the caching allocator's free pool filled with NaN / huge values -- every gradient must be bit-identical.
argv: [preset:gemm:B:text:frames ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from deepvoice3_pytorch_amd import builder, ops, train_step
dev = torch.device("cuda:0")


def poison(val, gb=8):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    t = torch.full((gb << 28,), val, device=dev)
    torch.cuda.synchronize()
    del t


def run(preset, gemm, B, Tt, frames, val):
    run_ = bench.TrainRun(dev, None, 0, 1, preset, gemm, B, Tt, frames, graph=False)
    tr = run_.trainer
    ops.dropout_state.manual_seed(777)
    ops.mask_plan.__init__()
    tr.arena.grad.zero_()
    if val is not None:
        poison(val)
    scal = tr.forward_backward(run_.batch)
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().clone() for k, p in tr.model.named_parameters() if p.grad is not None}
    run_.close()
    return {k: float(v) for k, v in scal.items()}, grads


cases = [c.split(":") for c in sys.argv[1:]] or [("deepvoice3_ljspeech", "f16x3", 4, 40, 120), ("deepvoice3_ljspeech", "f16x3", 16, 150, 800),
                                                 ("nyanko_ljspeech", "bf16", 4, 40, 120), ("deepvoice3_vctk", "bf16", 4, 40, 120),
                                                 ("deepvoice3_ljspeech", "f16x3", 64, 150, 800), ("nyanko_ljspeech", "bf16", 64, 150, 800)]
for preset, gemm, B, Tt, frames in cases:
    B, Tt, frames = int(B), int(Tt), int(frames)
    s0, g0 = run(preset, gemm, B, Tt, frames, None)
    for val in (float("nan"), 3.0e30):
        s1, g1 = run(preset, gemm, B, Tt, frames, val)
        diff = [k for k in g0 if not torch.equal(g0[k], g1[k])]
        print(preset, gemm, "B=%d T=%d/%d" % (B, Tt, frames), "pool filled with %r:" % val,
              "identical" if not diff and s0 == s1 else "DIFFERENT: %d gradients, first %s; loss %r vs %r" % (len(diff), diff[:6], s0.get("loss"), s1.get("loss")), flush=True)
