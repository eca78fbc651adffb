# coding: utf-8
"""Round 5, shipped library: the run-time selectable LOAD-phase structures.
(1) conv_gemm_pp2 (fp16-pair forward, bf16-pair input gradient): ORD 0 / 17 / 81 (dv3_debug_set(29, v) unmasked,
    (31, v) masked) bit-identical to each other, graph-timed at the north-star shape.
(2) conv_c8pp (single-term bf16 on c8 tensors): fragment reads first (dv3_debug_set(30, 1)) bit-identical to the
    staging-first order, graph-timed over the presets' shapes."""
import math
import sys
import torch
from r5_common import ops, L, dev, graph_time, north_star

i32 = lambda t: t.view(torch.int32) if t.dtype == torch.float32 else t.view(torch.int16)
VARS = (0, 17, 81)


def set_ord(o):
    L.dv3_debug_set(29, o)
    L.dv3_debug_set(31, o)


ok = True
for mode in ("f16x3", "bf16x3"):
    ops.set_gemm_precision(mode)
    for (B, C, T, d, causal, masked) in [(3, 64, 75, 2, False, False), (2, 256, 150, 27, False, False), (2, 128, 100, 1, True, True),
                                         (5, 96, 61, 9, False, True), (4, 256, 800, 3, False, True), (7, 32, 33, 1, False, False),
                                         (8, 256, 1024, 1, False, False), (8, 256, 1024, 27, True, True), (2, 64, 300, 1, False, False)]:
        x, bias, pk, bits, rs, kb = north_star(masked, C=C, B=B, T=T, zero_bias=False)
        if T == 300:
            x[0, 3, 17] = 5000.0
            x[1, 40, 200] = -9000.0
        padL = 2 * d if causal else d
        gm = torch.randn(B, 2 * C, T, device=dev)
        outs = []
        for o in VARS:
            y = torch.full((B, C, T), 7.0, device=dev)
            ab = torch.full((B, 2 * C, T), 7.0, device=dev)
            dx = torch.full((B, C, T), 7.0, device=dev)
            kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=3, dil=d, padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x, residual=1,
                      a_split=pk.fwd_s, xmask=bits, xmask_rs=rs or 0, xmask_c8=kb, drop_scale=1 / 0.95 if masked else 1.0,
                      tile_hint=30, y=y, ab=ab)
            dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=3, dil=d, padL=2 * d - padL, mode=ops.EPI_DGRAD, r=x, ymask=bits,
                       ymask_rs=rs or 0, drop_scale=1 / 0.95 if masked else 1.0, a_split=pk.bwd_s, tile_hint=30, y=dx)
            ops.f16_range_events(reset=True)
            set_ord(o)
            ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw)
            ops.conv_gemm(gm, None, pk.ldb, 0, **dkw)
            set_ord(0)
            outs.append((y, ab, dx, ops.f16_range_events(reset=True)))
        same = all(torch.equal(i32(outs[0][0]), i32(o_[0])) and torch.equal(i32(outs[0][1]), i32(o_[1])) and
                   torch.equal(i32(outs[0][2]), i32(o_[2])) and outs[0][3] == o_[3] for o_ in outs[1:])
        ok &= same
        print("%-6s B=%d C=%d T=%d d=%d causal=%d masked=%d: %s   range events %s" % (mode, B, C, T, d, causal, masked,
              "BIT-EQUAL" if same else "DIFFERS", [o_[3] for o_ in outs]), flush=True)
print("pp2: ALL BIT-EQUAL" if ok else "pp2: MISMATCH", flush=True)

B, C, T, k = 64, 256, 1024, 3
for mode in ("f16x3", "bf16x3"):
    ops.set_gemm_precision(mode)
    x, bias, pk, bits, rs, kb = north_star(True)
    y = torch.empty(B, C, T, device=dev)
    ab = torch.empty(B, 2 * C, T, device=dev)
    gm = torch.randn(B, 2 * C, T, device=dev)
    dx = torch.empty(B, C, T, device=dev)
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x, residual=1,
              a_split=pk.fwd_s, y=y, tile_hint=30)
    mkw = dict(kw, xmask=bits, xmask_rs=rs, xmask_c8=kb, drop_scale=1 / 0.95, ab=ab)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_DGRAD, ymask=bits, ymask_rs=rs,
               drop_scale=1 / 0.95, a_split=pk.bwd_s, y=dx, r=x, r_scale=0.7071, tile_hint=30)
    for rnd in range(3):
        for o in VARS:
            set_ord(o)
            te = graph_time(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw))
            tm = graph_time(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **mkw))
            td = graph_time(lambda: ops.conv_gemm(gm, None, pk.ldb, 0, **dkw))
            print("%-6s ORD %2d: eval fwd %.1f us   train fwd (masked, pre-gate save) %.1f us   dgrad %.1f us" % (mode, o, te, tm, td), flush=True)
    set_ord(0)

# ---- conv_c8pp: reads first ----
ops.set_gemm_precision("bf16")
ops.bf16_storage = True
B = 64
okc = True
shapes = [(256, 1024, 1, False, 3), (256, 1024, 27, False, 3), (512, 150, 1, False, 3), (256, 400, 3, False, 3), (256, 800, 1, False, 3),
          (512, 800, 3, False, 3), (512, 150, 1, False, 1), (256, 800, 1, False, 1), (96, 333, 9, True, 3)]
for (C, T, d, causal, k) in shapes:
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.randn(2 * C, device=dev) * 0.1
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True, split_only=True)
    x8 = ops.to_c8(x)
    ops.dropout_state.manual_seed(3)
    keep8 = ops.dropout_keep_c8(B, C, T, 0.05, dev)
    gm8 = ops.to_c8(torch.randn(B, 2 * C, T, device=dev))
    padL = (k - 1) * d if causal else (k - 1) // 2 * d
    L.dv3_debug_set(19, 1)
    res, outs = [], []
    for rf in (0, 1, 0, 1):
        L.dv3_debug_set(30, rf)
        y8 = ops._c8_empty(B, C, T, dev); ab = ops._c8_empty(B, 2 * C, T, dev); ym8 = ops._c8_empty(B, C, T, dev); dx8 = ops._c8_empty(B, C, T, dev)
        ekw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x8,
                   residual=1, a_split=pk.fwd_s, x_c8=x8, out_c8=True)
        mkw = dict(ekw, xmask_c8=keep8, drop_scale=1 / 0.95, ab=ab, y=ym8)
        dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=d, padL=(k - 1) * d - padL, mode=ops.EPI_DGRAD, r=x8, r_scale=0.7071,
                   drop_scale=1 / 0.95, a_split=pk.bwd_s, x_c8=gm8, out_c8=True, ymask_c8=keep8, y=dx8)
        ops.conv_gemm(None, None, pk.lda, pk.a_half, y=y8, **ekw)
        vf = L.dv3_debug_get(10)
        ops.conv_gemm(None, None, pk.lda, pk.a_half, **mkw)
        ops.conv_gemm(None, None, pk.ldb, 0, **dkw)
        torch.cuda.synchronize()
        outs.append((y8, ym8, ab, dx8))
        te = graph_time(lambda: ops.conv_gemm(None, None, pk.lda, pk.a_half, y=y8, **ekw))
        tm = graph_time(lambda: ops.conv_gemm(None, None, pk.lda, pk.a_half, **mkw))
        td = graph_time(lambda: ops.conv_gemm(None, None, pk.ldb, 0, **dkw))
        res.append((te, tm, td, vf))
    L.dv3_debug_set(30, 0)
    L.dv3_debug_set(19, 128)
    same = all(all(torch.equal(a_.view(torch.int16), b_.view(torch.int16)) for a_, b_ in zip(outs[0], o_)) for o_ in outs[1:])
    okc &= same
    fl = 2.0 * B * T * (2 * C) * (k * C)
    print("c8pp C=%3d T=%4d k=%d d=%2d causal=%d variant %d %s: eval %6.1f / %6.1f -> %6.1f / %6.1f us (%.0f TF)   train fwd %6.1f / %6.1f -> %6.1f / %6.1f   dgrad %6.1f / %6.1f -> %6.1f / %6.1f" % (
        C, T, k, d, causal, res[0][3], "BIT-EQUAL" if same else "DIFFERS", res[0][0], res[2][0], res[1][0], res[3][0], fl / min(res[1][0], res[3][0]) / 1e6,
        res[0][1], res[2][1], res[1][1], res[3][1], res[0][2], res[2][2], res[1][2], res[3][2]), flush=True)
print("c8pp: ALL BIT-EQUAL" if okc else "c8pp: MISMATCH", flush=True)
