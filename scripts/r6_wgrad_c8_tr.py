# coding: utf-8
"""Round 6: wgrad_c8 with operand fragments cut by ds_read_b64_tr_b16 from the untransposed tile (dv3_debug_set(52, 1))
against the register-transposing forms (52, 0), per launch from a hipGraph, over the bf16 presets' layer shapes: same bits?
how long?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from scripts.r5_common import graph_time, dev, L
from deepvoice3_pytorch_amd import ops

ops.set_gemm_precision("bf16")
ops.bf16_storage = True
SHAPES = ((64, 256, 512, 1024, 3, 1, True), (64, 256, 512, 804, 3, 1, True), (64, 512, 1024, 804, 3, 3, True), (64, 512, 1024, 150, 3, 1, True),
          (64, 512, 1024, 150, 3, 27, True), (64, 256, 512, 201, 3, 1, True), (64, 256, 512, 201, 3, 9, True), (64, 256, 512, 402, 3, 3, True),
          (64, 513, 513, 804, 1, 1, False), (64, 256, 256, 201, 1, 1, False), (64, 512, 512, 150, 1, 1, True), (16, 256, 512, 804, 3, 1, True),
          (3, 72, 136, 77, 3, 2, True), (2, 80, 256, 50, 1, 1, False))
for B, C, M, T, J, d, masked in SHAPES:
    torch.manual_seed(0)
    x8 = ops.to_c8(torch.randn(B, C, T, device=dev))
    g8 = ops.to_c8(torch.randn(B, M, T, device=dev))
    keep = None
    if masked:
        ops.dropout_state.manual_seed(3)
        keep = ops.dropout_keep_c8(B, C, T, 0.05, dev)
        keep = keep[0] if isinstance(keep, tuple) else keep
    tiles = ((M + 127) // 128) * ((C + 127) // 128)
    S = ops._ksplit_count(B * ((T + 31) // 32), tiles, slots=256, c8=True)
    res, ref = {}, None
    for tr in (0, 1, 2, 3, 4):
        L.dv3_debug_set(52, tr)
        f = lambda: ops.wgrad_gemm_c8(g8, x8, B=B, M=M, Cin=C, T=T, J=J, dil=d, padL=d * (J // 2), n_slabs=S, xmask_c8=keep,
                                      drop_scale=1 / 0.95 if masked else 1.0, rows_of_slabs=True)
        o = f()
        v = L.dv3_debug_get(11)
        if ref is None:
            ref = o.clone()
        res["%d" % v] = (graph_time(f), torch.equal(o, ref))
    L.dv3_debug_set(52, 1)
    fl = 2.0 * B * T * M * J * C
    print("B=%d C=%d M=%d T=%d J=%d d=%d S=%d:" % (B, C, M, T, J, d, S), "  ".join("%s %.1f us%s" % (k, t, "" if ok else " DIFFERS") for k, (t, ok) in res.items()),
          "| best = %.3f of 2.5 PF" % (fl / (min(t for t, _ in res.values()) * 1e-6) / 2.5e15), flush=True)
