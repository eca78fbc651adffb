# coding: utf-8
"""Round 6: the weight-gradient kernel (wgrad_taps2) per launch, hipGraph-timed: per-tap staging against the one-window
form (dv3_debug_set(47, 0 | 1)), fp32 against pair-word g, over the shapes of the presets' three-tap layers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from scripts.r5_common import graph_time, dev, L
from deepvoice3_pytorch_amd import ops

ops.set_gemm_precision("f16x3")
SHAPES = ((64, 256, 1024, 1), (64, 256, 1024, 3), (64, 256, 804, 1), (64, 256, 804, 3), (64, 512, 804, 1), (64, 512, 804, 3),
          (64, 256, 402, 1), (64, 512, 150, 1), (64, 512, 150, 3), (64, 256, 200, 1), (16, 256, 804, 1), (16, 512, 150, 3))
for B, C, T, d in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    g = torch.randn(B, 2 * C, T, device=dev)
    gp = ops.pair_words_of(g)
    ops.dropout_state.manual_seed(3)
    bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
    tiles = ((2 * C + 127) // 128) * ((C + 127) // 128)
    S = ops._ksplit_count(B * ((T + 31) // 32), tiles, slots=256)
    res, ref = {}, None
    for win, il in ((0, 0), (1, 0), (1, 1)):
        for pair in (False, True):
            L.dv3_debug_set(47, win)
            L.dv3_debug_set(48, il)
            f = lambda: ops.wgrad_gemm(gp if pair else g, x, B=B, M=2 * C, Cin=C, T=T, Tin=T, J=3, dil=d, padL=d, n_slabs=S,
                                       xmask=bits, xmask_rs=rs, drop_scale=1 / 0.95, split_bf16=True, k_split=True,
                                       rows_of_slabs=True, g_pair=pair)
            o = f()
            v = L.dv3_debug_get(11)
            if ref is None:
                ref = o.clone()
            same = torch.equal(o, ref)
            res["%s%s (%d)" % ("window, interleaved" if il else "window" if win else "per-tap", " + pair g" if pair else "", v)] = (graph_time(f), same)
    L.dv3_debug_set(47, 1)
    L.dv3_debug_set(48, 1)
    fl = 2.0 * B * T * 2 * C * 3 * C
    print("B=%d C=%d T=%d d=%d S=%d:" % (B, C, T, d, S),
          "  ".join("%s %.1f us%s" % (k, t, "" if ok else " DIFFERS") for k, (t, ok) in res.items()),
          "| best = %.3f of the 833 TF split roof" % (fl / (min(t for t, _ in res.values()) * 1e-6) / 833e12), flush=True)
