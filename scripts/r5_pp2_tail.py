# coding: utf-8
"""Round 5: the Conv1dGLU tail of the 256 x 256 tap-GEMM.  (1) bit-identity of the 16-byte quad-transpose tail (ORD 32)
with the shipped 4-byte tail, incl. the pre-gate save, ragged tile edges and the highway form; (2) graph-timed launches
(not host-bound) of the shipped kernel, the ORD variants and the tail ablations: what the tail costs and whether loads or
stores bound it.  Experiment build (DV3_LIBPATH=.../libdv3hip_exp.so)."""
import torch
from r5_common import ops, L, dev, graph_time, north_star

ops.set_gemm_precision("f16x3")


def run(o, x, pk, kw):
    L.dv3_debug_set(29, o)
    ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw)
    L.dv3_debug_set(29, 0)


ok = True
for (B, C, T, d, causal, masked, mode) in [(3, 64, 76, 2, False, False, ops.EPI_GLU), (2, 256, 152, 27, False, False, ops.EPI_GLU),
                                           (2, 128, 100, 1, True, True, ops.EPI_GLU), (5, 96, 64, 9, False, True, ops.EPI_GLU),
                                           (4, 256, 800, 3, False, True, ops.EPI_GLU), (7, 32, 36, 1, False, False, ops.EPI_HIGHWAY),
                                           (8, 256, 1024, 1, False, False, ops.EPI_GLU), (3, 64, 75, 2, False, False, ops.EPI_GLU)]:
    x, bias, pk, bits, rs, kb = north_star(masked, C=C, B=B, T=T, zero_bias=False)
    padL = 2 * d if causal else d
    outs = []
    for o in (0, 32, 49):
        y = torch.full((B, C, T), 7.0, device=dev)
        ab = torch.full((B, 2 * C, T), 7.0, device=dev)
        kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=3, dil=d, padL=padL, mode=mode, Cg=C, bias=bias, r=x, residual=1,
                  a_split=pk.fwd_s, xmask=bits, xmask_rs=rs or 0, xmask_c8=kb, drop_scale=1 / 0.95 if masked else 1.0,
                  tile_hint=30, y=y, ab=ab)
        run(o, x, pk, kw)
        torch.cuda.synchronize()
        outs.append((y, ab))
    same = all(torch.equal(outs[0][0].view(torch.int32), o_[0].view(torch.int32)) and
               torch.equal(outs[0][1].view(torch.int32), o_[1].view(torch.int32)) for o_ in outs[1:])
    ok &= same
    print("B=%d C=%d T=%d d=%d causal=%d masked=%d mode=%d: 16-byte tail %s (max diff y %.2e ab %.2e)" % (
        B, C, T, d, causal, masked, mode, "BIT-EQUAL" if same else "DIFFERS",
        max(float((outs[0][0] - o_[0]).abs().max()) for o_ in outs[1:]), max(float((outs[0][1] - o_[1]).abs().max()) for o_ in outs[1:])), flush=True)
print("ALL BIT-EQUAL" if ok else "MISMATCH", flush=True)

# ---- timing ----
B, C, T, k = 64, 256, 1024, 3
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
    for o in (0, 1, 17, 32, 49):
        L.dv3_debug_set(29, o)
        te = graph_time(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw))
        tm = graph_time(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **mkw)) if o != 1 else float("nan")
        td = graph_time(lambda: ops.conv_gemm(gm, None, pk.ldb, 0, **dkw))
        print("ORD %2d: eval fwd %.1f us   train fwd (masked, pre-gate save) %.1f us   dgrad %.1f us" % (o, te, tm, td), flush=True)
L.dv3_debug_set(29, 0)
for rnd in range(2):
    for abl, name in ((0, "full"), (1, "no MFMAs"), (2, "no staging"), (3, "no tail"), (12, "4-byte tail, no residual load"),
                      (13, "4-byte tail, no stores"), (14, "16-byte tail, no residual load"), (15, "16-byte tail, no stores")):
        L.dv3_debug_set(13, abl)
        print("ablation %-34s: %.1f us" % (name, graph_time(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw))), flush=True)
L.dv3_debug_set(13, 0)
# same instruction stream, operands that toggle fewer multiplier bits (the power budget): zero activations
xz = torch.zeros_like(x)
kwz = dict(kw, r=xz)
for rnd in range(2):
    print("zero activations: %.1f us" % graph_time(lambda: ops.conv_gemm(xz, None, pk.lda, pk.a_half, **kwz)), flush=True)
    for abl, name in ((1, "no MFMAs"), (2, "no staging"), (3, "no tail")):
        L.dv3_debug_set(13, abl)
        print("zero activations, ablation %-12s: %.1f us" % (name, graph_time(lambda: ops.conv_gemm(xz, None, pk.lda, pk.a_half, **kwz))), flush=True)
    L.dv3_debug_set(13, 0)
