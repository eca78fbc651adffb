# coding: utf-8
"""Round 6: the 256 x 256 tap-GEMM's gated tail: guarded per element (dv3_debug_set(50, 0)) against the straight-line
tail of interior sub-tiles (50, 1: conv_common.h conv_epilogue_glu_interior).  Bit-identity over edge shapes (rows /
columns that leave the tensor, highway, no residual, masked with the pre-gate save), then graph-timed at the
north-star shape and the converter's shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
from r5_common import ops, L, dev, graph_time, north_star

i32 = lambda t: t.view(torch.int32)
ok = True
for mode in ("f16x3", "bf16x3"):
    ops.set_gemm_precision(mode)
    for (B, C, T, d, causal, masked, epi, res) in [(3, 64, 75, 2, False, False, "glu", 1), (2, 256, 150, 27, False, True, "glu", 1),
                                                   (2, 128, 100, 1, True, True, "hw", 0), (5, 96, 61, 9, False, True, "glu", 0),
                                                   (4, 256, 800, 3, False, True, "glu", 1), (8, 256, 1024, 1, False, False, "hw", 0),
                                                   (8, 256, 1000, 27, True, True, "glu", 1), (3, 320, 300, 1, False, False, "glu", 1)]:
        x, bias, pk, bits, rs, kb = north_star(masked, C=C, B=B, T=T, zero_bias=False)
        padL = 2 * d if causal else d
        outs = []
        for v in (0, 1):
            y = torch.full((B, C, T), 7.0, device=dev)
            ab = torch.full((B, 2 * C, T), 7.0, device=dev) if masked else None
            L.dv3_debug_set(50, v)
            ops.conv_gemm(x, None, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=3, dil=d, padL=padL,
                          mode=ops.EPI_GLU if epi == "glu" else ops.EPI_HIGHWAY, Cg=C, bias=bias, r=x, residual=res,
                          a_split=pk.fwd_s, xmask=bits, xmask_rs=rs or 0, xmask_c8=kb, drop_scale=1 / 0.95 if masked else 1.0,
                          tile_hint=30, y=y, ab=ab)
            outs.append((y, ab))
        same = torch.equal(i32(outs[0][0]), i32(outs[1][0])) and (not masked or torch.equal(i32(outs[0][1]), i32(outs[1][1])))
        ok &= same
        print("%-6s B=%d C=%d T=%d d=%d causal=%d masked=%d %s res=%d: %s" % (mode, B, C, T, d, causal, masked, epi, res,
              "BIT-EQUAL" if same else "DIFFERS"), flush=True)
L.dv3_debug_set(50, 1)
print("fast tail: ALL BIT-EQUAL" if ok else "fast tail: MISMATCH", flush=True)

ops.set_gemm_precision("f16x3")
for (B, C, T) in ((64, 256, 1024), (64, 256, 804), (64, 512, 804)):
    x, bias, pk, bits, rs, kb = north_star(True, C=C, B=B, T=T)
    y = torch.empty(B, C, T, device=dev)
    ab = torch.empty(B, 2 * C, T, device=dev)
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=3, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x, residual=1,
              a_split=pk.fwd_s, y=y)
    mkw = dict(kw, xmask=bits, xmask_rs=rs, xmask_c8=kb, drop_scale=1 / 0.95, ab=ab)
    res = {}
    for rnd in range(3):
        for v in (0, 1):
            L.dv3_debug_set(50, v)
            res.setdefault(("eval forward", v), []).append(round(graph_time(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw), per_graph=25), 1))
            res.setdefault(("masked training forward + pre-gate save", v), []).append(round(graph_time(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, **mkw), per_graph=25), 1))
    L.dv3_debug_set(50, 1)
    print("B=%d C=%d T=%d (variant %d):" % (B, C, T, L.dv3_debug_get(10)),
          " | ".join("%s %s: %s us" % (k, "straight-line tail" if v else "guarded tail", t) for (k, v), t in res.items()), flush=True)
