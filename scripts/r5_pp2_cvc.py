# coding: utf-8
"""Round 5: activation conversion inside COMPUTE phases (ORD 64): bit-identity (incl. the range counter) and graph timing
on random and zero activations."""
import torch
from r5_common import ops, L, dev, graph_time, north_star
ops.set_gemm_precision("f16x3")
VARS = (0, 1, 17, 64, 65, 81)
ok = True
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
        L.dv3_debug_set(29, o)
        ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw)
        ops.conv_gemm(gm, None, pk.ldb, 0, **dkw)
        L.dv3_debug_set(29, 0)
        ev = ops.f16_range_events(reset=True)
        outs.append((y, ab, dx, ev))
    i32 = lambda t: t.view(torch.int32)
    same = all(torch.equal(i32(outs[0][0]), i32(o_[0])) and torch.equal(i32(outs[0][1]), i32(o_[1])) and
               torch.equal(i32(outs[0][2]), i32(o_[2])) and outs[0][3] == o_[3] for o_ in outs[1:])
    ok &= same
    print("B=%d C=%d T=%d d=%d causal=%d masked=%d: %s   range events %s" % (B, C, T, d, causal, masked,
          "BIT-EQUAL" if same else "DIFFERS", [o_[3] for o_ in outs]), flush=True)
print("ALL BIT-EQUAL" if ok else "MISMATCH", flush=True)

B, C, T, k = 64, 256, 1024, 3
x, bias, pk, bits, rs, kb = north_star(True)
y = torch.empty(B, C, T, device=dev)
ab = torch.empty(B, 2 * C, T, device=dev)
gm = torch.randn(B, 2 * C, T, device=dev)
dx = torch.empty(B, C, T, device=dev)
xz = torch.zeros_like(x)
for name, xx in (("randn", x), ("zero", xz)):
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C, bias=bias, r=xx, residual=1,
              a_split=pk.fwd_s, y=y, tile_hint=30)
    mkw = dict(kw, xmask=bits, xmask_rs=rs, xmask_c8=kb, drop_scale=1 / 0.95, ab=ab)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_DGRAD, ymask=bits, ymask_rs=rs,
               drop_scale=1 / 0.95, a_split=pk.bwd_s, y=dx, r=xx, r_scale=0.7071, tile_hint=30)
    gmx = gm if name == "randn" else torch.zeros_like(gm)
    for rnd in range(3):
        for o in VARS:
            L.dv3_debug_set(29, o)
            te = graph_time(lambda: ops.conv_gemm(xx, None, pk.lda, pk.a_half, **kw))
            tm = graph_time(lambda: ops.conv_gemm(xx, None, pk.lda, pk.a_half, **mkw)) if o != 1 else float("nan")
            td = graph_time(lambda: ops.conv_gemm(gmx, None, pk.ldb, 0, **dkw))
            print("x %-5s ORD %2d: eval fwd %.1f us   train fwd (masked, pre-gate save) %.1f us   dgrad %.1f us" % (name, o, te, tm, td), flush=True)
    L.dv3_debug_set(29, 0)
