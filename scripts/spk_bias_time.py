# coding: utf-8
"""Time of the fused speaker-bias block kernels (csrc/speaker_bias.hip) at the block shapes of deepvoice3_vctk (B = 64),
with timing-only ablations of the backward's first pass (dv3_debug_set(28, v): 1 no dW, 2 no d emb, 3 no loads)."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepvoice3_pytorch_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()


def timeit(fn, iters=10, settle=4):
    for _ in range(settle):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


B, E = 64, 16
for (T, Cs) in [(201, (256,) * 10), (150, (512,) * 7), (804, (256,) * 6)]:
    torch.manual_seed(0)
    e = torch.randn(B, E, T, device=dev, requires_grad=True)
    layers = [(torch.randn(C, E, device=dev, requires_grad=True), torch.rand(C, 1, device=dev).add_(0.5).requires_grad_(True),
               torch.randn(C, device=dev).requires_grad_(True)) for C in Cs]
    douts = [torch.randn(B, C, T, device=dev) for C in Cs]
    tf = timeit(lambda: ops.speaker_bias_block(e.detach(), layers))
    outs = ops.speaker_bias_block(e, layers)
    res = []
    for abl in (0, 1, 2, 3):
        L.dv3_debug_set(28, abl)
        res.append("abl %d: %.1f" % (abl, timeit(lambda: torch.autograd.backward(outs, douts, retain_graph=True))))
    L.dv3_debug_set(28, 0)
    mb = sum(B * C * T * 4 for C in Cs) / 1e6
    print("T=%d layers=%d C=%d (%.0f MB of biases): forward %.1f us | backward (3 launches) %s" % (T, len(Cs), Cs[0], mb, tf, "  ".join(res)), flush=True)
