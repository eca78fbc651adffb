# coding: utf-8
"""Two-steps-ahead all-taps wgrad (csrc/wgrad_taps2.hip, dv3_debug_set(2, 4)): bit-equality of the slabs with the shipped all-taps
kernel, timing and ablations at the north-star shape."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepvoice3_pytorch_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()


def timeit(fn, iters=30, settle=30):
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


def run(B, M, C, T, d, masked, S, mode):
    ops.set_gemm_precision(mode)
    torch.manual_seed(1)
    x = torch.randn(B, C, T, device=dev)
    g = torch.randn(B, M, T, device=dev)
    bits = rs = None
    if masked:
        ops.dropout_state.manual_seed(3)
        bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
    outs = []
    for tile in (3, 4):
        L.dv3_debug_set(2, tile)
        o = ops.wgrad_gemm(g, x, B=B, M=M, Cin=C, T=T, Tin=T, J=3, dil=d, padL=d, n_slabs=S, xmask=bits,
                           xmask_rs=rs or 0, drop_scale=1 / 0.95 if masked else 1.0, split_bf16=True, k_split=True)
        outs.append((o, L.dv3_debug_get(11)))
    L.dv3_debug_set(2, 0)
    ok = torch.equal(outs[0][0], outs[1][0])
    print("%s B=%d M=%d C=%d T=%d d=%d masked=%d S=%d variants %d/%d: %s (max diff %.2e)" % (
        mode, B, M, C, T, d, masked, S, outs[0][1], outs[1][1], "BIT-EQUAL" if ok else "DIFFERS",
        float((outs[0][0] - outs[1][0]).abs().max())))
    return ok


ok = True
for mode in ("f16x3", "bf16x3"):
    for (B, M, C, T, d, masked, S) in [(3, 128, 64, 75, 2, False, 2), (2, 512, 256, 150, 27, True, 3), (4, 96, 200, 61, 9, True, 5),
                                       (5, 256, 128, 800, 1, True, 7), (2, 130, 130, 33, 3, True, 2),
                                       (6, 72, 64, 100, 1, False, 19)]:
        ok &= run(B, M, C, T, d, masked, S, mode)
print("ALL BIT-EQUAL" if ok else "MISMATCH")

ops.set_gemm_precision("f16x3")
B, C, T, k = 64, 256, 1024, 3
torch.manual_seed(0)
x = torch.randn(B, C, T, device=dev)
gm = torch.randn(B, 2 * C, T, device=dev)
bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
S = ops._ksplit_count(B * ((T + 31) // 32), 8, slots=256)
out_t = torch.empty((S, k, 2 * C, C), dtype=torch.float32, device=dev)
for dil in (1, 27):
    for rnd in range(2):
        for tile in (3, 4):
            L.dv3_debug_set(2, tile)
            tm = timeit(lambda: ops.wgrad_gemm(gm, x, B=B, M=2 * C, Cin=C, T=T, Tin=T, J=k, dil=dil, padL=dil, n_slabs=S, xmask=bits,
                                               xmask_rs=rs, drop_scale=1 / 0.95, split_bf16=True, k_split=True, out=out_t))
            tu = timeit(lambda: ops.wgrad_gemm(gm, x, B=B, M=2 * C, Cin=C, T=T, Tin=T, J=k, dil=dil, padL=dil, n_slabs=S,
                                               split_bf16=True, k_split=True, out=out_t))
            print("dil %2d tile %d (%d slabs): masked %.1f us   unmasked %.1f us" % (dil, tile, S, tm, tu))
L.dv3_debug_set(2, 4)
for abl, name in ((0, "full"), (1, "no MFMAs"), (2, "no staging"), (6, "k16 blocks not pinned apart")):
    L.dv3_debug_set(16, abl)
    print("ablation %-20s: %.1f us" % (name, timeit(lambda: ops.wgrad_gemm(gm, x, B=B, M=2 * C, Cin=C, T=T, Tin=T, J=k, dil=1, padL=1,
                                                                            n_slabs=S, split_bf16=True, k_split=True, out=out_t))))
L.dv3_debug_set(16, 0)
L.dv3_debug_set(2, 0)
