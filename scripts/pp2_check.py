# coding: utf-8
"""256 x 256 k16 ping-pong tap-GEMM (csrc/conv_gemm_pp2.hip, tile_hint 30): bit-equality with the shipped kernels
(forward with pre-gate save, masked and not; input-gradient form) and timing / ablations at the north-star shape."""
import math
import sys
import os
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepvoice3_pytorch_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()


def timeit(fn, iters=40, settle=40):
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


def check(B, C, T, d, causal, masked, mode, k=3):
    ops.set_gemm_precision(mode)
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.randn(2 * C, device=dev) * 0.1
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True)
    bits = rs = kb = None
    if masked:
        ops.dropout_state.manual_seed(3)
        bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
        kb = ops.mask_bits_to_c8(bits, rs, B, C, T)
    padL = (k - 1) * d if causal else d
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
              residual=1, a_split=pk.fwd_s, xmask=bits, xmask_rs=rs or 0, xmask_c8=kb,
              drop_scale=1 / 0.95 if masked else 1.0)
    ys, abs_, var = [], [], []
    for hint in (0, 30):
        y = torch.empty(B, C, T, device=dev)
        ab = torch.empty(B, 2 * C, T, device=dev)
        ops.conv_gemm(x, None, pk.lda, pk.a_half, y=y, ab=ab, tile_hint=hint, **kw)
        var.append(L.dv3_debug_get(10))
        ys.append(y)
        abs_.append(ab)
    ok_f = torch.equal(ys[0], ys[1]) and torch.equal(abs_[0], abs_[1])
    gm = torch.randn(B, 2 * C, T, device=dev)
    dres = torch.randn(B, C, T, device=dev)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=d, padL=(k - 1) * d - padL, mode=ops.EPI_DGRAD,
               r=dres, ymask=bits, ymask_rs=rs or 0, drop_scale=1 / 0.95 if masked else 1.0, a_split=pk.bwd_s)
    dxs = []
    for hint in (0, 30):
        dx = torch.empty(B, C, T, device=dev)
        ops.conv_gemm(gm, None, pk.ldb, 0, y=dx, tile_hint=hint, **dkw)
        dxs.append(dx)
    ok_d = torch.equal(dxs[0], dxs[1])
    md = float((ys[0] - ys[1]).abs().max())
    print("%-6s B=%d C=%d T=%d d=%d causal=%d masked=%d variants %s: fwd %s (max diff %.2e) dgrad %s" % (
        mode, B, C, T, d, causal, masked, var, "BIT-EQUAL" if ok_f else "DIFFERS", md, "BIT-EQUAL" if ok_d else "DIFFERS"))
    return ok_f and ok_d


ok = True
for mode in ("f16x3", "bf16x3"):
    for (B, C, T, d, causal, masked) in [(3, 64, 75, 2, False, False), (2, 256, 150, 27, False, False),
                                         (2, 128, 100, 1, True, True), (5, 96, 61, 9, False, True),
                                         (4, 256, 800, 3, False, True), (7, 32, 33, 1, False, False)]:
        ok &= check(B, C, T, d, causal, masked, mode)
print("ALL BIT-EQUAL" if ok else "MISMATCH")

# ---- timing at the north-star shape ----
ops.set_gemm_precision("f16x3")
B, C, T, k = 64, 256, 1024, 3
torch.manual_seed(0)
x = torch.randn(B, C, T, device=dev)
v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
bias = torch.zeros(2 * C, device=dev)
pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True)
y = torch.empty(B, C, T, device=dev)
ab = torch.empty(B, 2 * C, T, device=dev)
ops.dropout_state.manual_seed(3)
bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
kb = ops.mask_bits_to_c8(bits, rs, B, C, T)
gm = torch.randn(B, 2 * C, T, device=dev)
dx = torch.empty(B, C, T, device=dev)
for dil in (1, 27):
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=dil, padL=dil, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
              residual=1, a_split=pk.fwd_s, y=y)
    mkw = dict(kw, xmask=bits, xmask_rs=rs, xmask_c8=kb, drop_scale=1 / 0.95, ab=ab)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=dil, padL=(k - 1) * dil - dil, mode=ops.EPI_DGRAD,
               ymask=bits, ymask_rs=rs, drop_scale=1 / 0.95, a_split=pk.bwd_s, y=dx, r=x, r_scale=0.7071)
    for rnd in range(2):
        for hint in (0, 30):
            te = timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, tile_hint=hint, **kw))
            tm = timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, tile_hint=hint, **mkw))
            td = timeit(lambda: ops.conv_gemm(gm, None, pk.ldb, 0, tile_hint=hint, **dkw))
            print("dil %2d hint %2d: eval fwd %.1f us   train fwd (masked, pre-gate save) %.1f us   dgrad %.1f us" % (dil, hint, te, tm, td))
for wide in (0, 1, 0, 1):
    L.dv3_debug_set(18, wide)
    for hint in (29, 30):
        te = timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, tile_hint=hint, **kw))
        tm = timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, tile_hint=hint, **mkw))
        td = timeit(lambda: ops.conv_gemm(gm, None, pk.ldb, 0, tile_hint=hint, **dkw))
        print("wide epilogue %d hint %2d: eval fwd %.1f us   train fwd %.1f us   dgrad %.1f us" % (wide, hint, te, tm, td))
L.dv3_debug_set(18, 1)
kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
          residual=1, a_split=pk.fwd_s, y=y, tile_hint=30)
for abl, name in ((0, "full"), (5, "MFMAs not pinned between the barriers"), (1, "no MFMAs"), (2, "no staging"), (3, "no tail"),
                  (6, "no activation fetches"), (7, "no panel fetches"), (8, "no activation conversion / stores"),
                  (9, "no panel stores"), (10, "narrow (4-byte) tail")):
    L.dv3_debug_set(13, abl)
    print("ablation %-38s: %.1f us" % (name, timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw))))
L.dv3_debug_set(13, 0)
