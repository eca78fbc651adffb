# coding: utf-8
"""Round 6: wgrad_c8 (bf16 storage, three taps) per launch, hipGraph-timed: staging after the MFMAs (dv3_debug_set(49, 0))
against between them (49, 1), over the shapes of the bf16 presets' three-tap layers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from scripts.r5_common import graph_time, dev, L
from deepvoice3_pytorch_amd import ops

ops.set_gemm_precision("bf16")
ops.bf16_storage = True
for B, C, T, d in ((64, 256, 1024, 1), (64, 256, 804, 1), (64, 512, 804, 3), (64, 512, 150, 3), (64, 256, 200, 1), (64, 256, 402, 3), (16, 256, 804, 1)):
    torch.manual_seed(0)
    x8 = ops.to_c8(torch.randn(B, C, T, device=dev))
    g8 = ops.to_c8(torch.randn(B, 2 * C, T, device=dev))
    ops.dropout_state.manual_seed(3)
    keep = ops.dropout_keep_c8(B, C, T, 0.05, dev)
    keep = keep[0] if isinstance(keep, tuple) else keep
    tiles = ((2 * C + 127) // 128) * ((C + 127) // 128)
    S = ops._ksplit_count(B * ((T + 31) // 32), tiles, slots=256)
    res, ref = {}, None
    for il in (0, 1):
        L.dv3_debug_set(49, il)
        f = lambda: ops.wgrad_gemm_c8(g8, x8, B=B, M=2 * C, Cin=C, T=T, J=3, dil=d, padL=d, n_slabs=S, xmask_c8=keep,
                                      drop_scale=1 / 0.95, rows_of_slabs=True)
        o = f()
        v = L.dv3_debug_get(11)
        if ref is None:
            ref = o.clone()
        res["%s (%d)" % ("between the MFMAs" if il else "after the MFMAs", v)] = (graph_time(f), torch.equal(o, ref))
    L.dv3_debug_set(49, 1)
    fl = 2.0 * B * T * 2 * C * 3 * C
    print("B=%d C=%d T=%d d=%d S=%d:" % (B, C, T, d, S), "  ".join("%s %.1f us%s" % (k, t, "" if ok else " DIFFERS") for k, (t, ok) in res.items()),
          "| best = %.3f of 2.5 PF" % (fl / (min(t for t, _ in res.values()) * 1e-6) / 2.5e15), flush=True)
