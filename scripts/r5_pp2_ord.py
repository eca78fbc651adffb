# coding: utf-8
"""Round 5: LOAD-phase structure variants of the 256 x 256 k16 ping-pong tap-GEMM (conv_gemm_pp2.hip, template ORD,
dv3_debug_set(29, v); experiment build: make EXP=1, DV3_LIBPATH=.../libdv3hip_exp.so).

ORD bits: 1 fragment reads first, 2 lean staging (panel offsets in registers, clamp-free fp16 pair on the in-range
path), 4 residual rows touched during the last chunk, 8 residual of the pair by v_fma_mix, 16 only the weight fragments
first.  Every variant must be BIT-IDENTICAL to the shipped kernel (same operands, same accumulation order); timing at the
north-star shape, variants interleaved over several rounds in one process."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepvoice3_pytorch_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()

UNMASKED = [0, 1, 2, 3, 4, 5, 7, 10, 11, 15, 17, 19, 27, 31]
MASKED = [0, 17, 19, 27, 31]


def timeit(fn, iters=30, settle=10):
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


def make(B, C, T, k, masked):
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
    return x, bias, pk, bits, rs, kb


def check(B, C, T, d, causal, masked, k=3):
    x, bias, pk, bits, rs, kb = make(B, C, T, k, masked)
    padL = (k - 1) * d if causal else d
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
              residual=1, a_split=pk.fwd_s, xmask=bits, xmask_rs=rs or 0, xmask_c8=kb,
              drop_scale=1 / 0.95 if masked else 1.0, tile_hint=30)
    gm = torch.randn(B, 2 * C, T, device=dev)
    dres = torch.randn(B, C, T, device=dev)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=d, padL=(k - 1) * d - padL, mode=ops.EPI_DGRAD,
               r=dres, ymask=bits, ymask_rs=rs or 0, drop_scale=1 / 0.95 if masked else 1.0, a_split=pk.bwd_s,
               tile_hint=30)
    ref = None
    bad = []
    for o in (MASKED if masked else UNMASKED):
        L.dv3_debug_set(29, o)
        y = torch.empty(B, C, T, device=dev)
        ab = torch.empty(B, 2 * C, T, device=dev)
        dx = torch.empty(B, C, T, device=dev)
        ops.conv_gemm(x, None, pk.lda, pk.a_half, y=y, ab=ab, **kw)
        assert L.dv3_debug_get(10) == 5101, L.dv3_debug_get(10)
        ops.conv_gemm(gm, None, pk.ldb, 0, y=dx, **dkw)
        torch.cuda.synchronize()
        if ref is None:
            ref = (y, ab, dx)
        else:
            same = torch.equal(y, ref[0]) and torch.equal(ab, ref[1]) and torch.equal(dx, ref[2])
            if not same:
                bad.append((o, float((y - ref[0]).abs().max()), float((ab - ref[1]).abs().max()), float((dx - ref[2]).abs().max())))
    L.dv3_debug_set(29, 0)
    print("B=%d C=%d T=%d d=%d causal=%d masked=%d: %s" % (B, C, T, d, causal, masked, "all BIT-EQUAL" if not bad else "DIFFER %s" % bad), flush=True)
    return not bad


ops.set_gemm_precision("f16x3")
ok = True
for (B, C, T, d, causal, masked) in [(3, 64, 75, 2, False, False), (2, 256, 150, 27, False, False),
                                     (2, 128, 100, 1, True, True), (5, 96, 61, 9, False, True),
                                     (4, 256, 800, 3, False, True), (7, 32, 33, 1, False, False),
                                     (8, 256, 1024, 1, False, False), (8, 256, 1024, 27, True, True)]:
    ok &= check(B, C, T, d, causal, masked)
# out-of-range values take the clamped path of the lean conversion: same pair as the shipped kernel
x, bias, pk, bits, rs, kb = make(2, 64, 300, 3, False)
x[0, 3, 17] = 5000.0
x[1, 40, 200] = -9000.0
kw = dict(B=2, Cin=64, Tin=300, M=128, Tout=300, J=3, dil=1, padL=1, mode=ops.EPI_GLU, Cg=64, bias=bias, r=x, residual=1,
          a_split=pk.fwd_s, tile_hint=30)
outs = []
for o in (0, 3, 11):
    L.dv3_debug_set(29, o)
    y = torch.empty(2, 64, 300, device=dev)
    ops.conv_gemm(x, None, pk.lda, pk.a_half, y=y, **kw)
    outs.append(y)
L.dv3_debug_set(29, 0)
same = all(torch.equal(outs[0], o_) for o_ in outs[1:])
print("out-of-range inputs: %s, range events %d" % ("BIT-EQUAL" if same else "DIFFER", ops.f16_range_events() if hasattr(ops, "f16_range_events") else -1))
ok &= same
print("ALL BIT-EQUAL" if ok else "MISMATCH", flush=True)

# ---- timing at the north-star shape ----
B, C, T, k = 64, 256, 1024, 3
x, bias, pk, bits, rs, kb = make(B, C, T, k, True)
bias = torch.zeros(2 * C, device=dev)
y = torch.empty(B, C, T, device=dev)
ab = torch.empty(B, 2 * C, T, device=dev)
gm = torch.randn(B, 2 * C, T, device=dev)
dx = torch.empty(B, C, T, device=dev)
res = {}
for dil in (1, 27):
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=dil, padL=dil, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
              residual=1, a_split=pk.fwd_s, y=y, tile_hint=30)
    mkw = dict(kw, xmask=bits, xmask_rs=rs, xmask_c8=kb, drop_scale=1 / 0.95, ab=ab)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=dil, padL=(k - 1) * dil - dil, mode=ops.EPI_DGRAD,
               ymask=bits, ymask_rs=rs, drop_scale=1 / 0.95, a_split=pk.bwd_s, y=dx, r=x, r_scale=0.7071, tile_hint=30)
    for rnd in range(3 if dil == 1 else 1):
        for o in UNMASKED:
            L.dv3_debug_set(29, o)
            te = timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw))
            td = timeit(lambda: ops.conv_gemm(gm, None, pk.ldb, 0, **dkw))
            tm = timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **mkw)) if o in MASKED else float("nan")
            res.setdefault((dil, o), []).append((te, tm, td))
            print("dil %2d ORD %2d: eval fwd %.1f us   train fwd (masked, pre-gate save) %.1f us   dgrad %.1f us" % (dil, o, te, tm, td), flush=True)
L.dv3_debug_set(29, 0)
print("---- best of rounds ----")
for (dil, o), v in sorted(res.items()):
    print("dil %2d ORD %2d: eval fwd %.1f   train fwd %.1f   dgrad %.1f" % (dil, o, min(a[0] for a in v), min(a[1] for a in v), min(a[2] for a in v)))
