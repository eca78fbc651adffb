# coding: utf-8
"""Round 5: timing-only ablations of the 256 x 256 tap-GEMM's staging on ZERO activations (no matrix-pipe power: the
clock stays up, what is left is the schedule's own latency) and on random ones; graph-timed."""
import torch
from r5_common import ops, L, dev, graph_time, north_star
ops.set_gemm_precision("f16x3")
B, C, T, k = 64, 256, 1024, 3
x, bias, pk, bits, rs, kb = north_star(False)
y = torch.empty(B, C, T, device=dev)
xz = torch.zeros_like(x)
for name, xx in (("zero", xz), ("randn", x)):
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C, bias=bias, r=xx, residual=1,
              a_split=pk.fwd_s, y=y, tile_hint=30)
    for rnd in range(2):
        for abl, an in ((0, "full"), (1, "no MFMAs"), (2, "no staging"), (3, "no tail"), (6, "no activation fetches"), (7, "no panel fetches"),
                        (8, "no activation conversion / stores"), (9, "no panel stores"), (5, "MFMAs not pinned")):
            L.dv3_debug_set(13, abl)
            print("x %-5s ablation %-34s: %.1f us" % (name, an, graph_time(lambda: ops.conv_gemm(xx, None, pk.lda, pk.a_half, **kw))), flush=True)
        L.dv3_debug_set(13, 0)
        for o in (1, 2, 3, 11, 17, 4):
            L.dv3_debug_set(29, o)
            print("x %-5s ORD %2d: %.1f us" % (name, o, graph_time(lambda: ops.conv_gemm(xx, None, pk.lda, pk.a_half, **kw))), flush=True)
        L.dv3_debug_set(29, 0)
