# coding: utf-8
"""Shared by the round-5 kernel scripts: a launch timer that is not bound by the host (ctypes issue costs ~100 us per
call on a slow box: any kernel shorter than that measured as the issue rate) -- N launches captured into one hipGraph,
replayed, timed with events around the replays."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepvoice3_pytorch_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()


def graph_time(fn, per_graph=20, replays=6, settle=2):
    """us per call of fn, from `replays` replays of a graph of `per_graph` calls"""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(per_graph):
                fn()
    torch.cuda.synchronize()
    for _ in range(settle):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (replays * per_graph)


def north_star(masked=True, seed=0, C=256, B=64, T=1024, k=3, zero_bias=True):
    torch.manual_seed(seed)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.zeros(2 * C, device=dev) if zero_bias else torch.randn(2 * C, device=dev) * 0.1
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True)
    bits = rs = kb = None
    if masked:
        ops.dropout_state.manual_seed(3)
        bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
        kb = ops.mask_bits_to_c8(bits, rs, B, C, T)
    return x, bias, pk, bits, rs, kb
