# coding: utf-8
"""Round 6: the input gradient of a gated layer (pair-word operand, keep-bit mask on the output side, skip-path addend)
per launch over the forced tiles of the split kernels (tile_hint 21..29 = the 128-wide kernel's tiles, 30 = the
256 x 256 kernel), graph-timed.  argv: B C T d ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import math
import torch
from scripts.r5_common import graph_time, dev, L
from deepvoice3_pytorch_amd import ops

ops.set_gemm_precision("f16x3")
shapes = [(64, 512, 150, 1), (64, 512, 150, 27), (64, 256, 200, 1), (64, 256, 402, 3), (64, 256, 804, 1), (16, 512, 150, 3)]
if len(sys.argv) > 4:
    shapes = [tuple(int(a) for a in sys.argv[1:5])]
for B, C, T, d in shapes:
    torch.manual_seed(0)
    k = 3
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True)
    dy = torch.randn(B, C, T, device=dev)
    dab = ops.pair_words_of(torch.randn(B, 2 * C, T, device=dev))
    ops.dropout_state.manual_seed(3)
    bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
    res = {}
    ref = None
    for hint in (0, 21, 22, 28, 29, 30):
        def f():
            return ops.conv_gemm(dab, pk.bwd, pk.ldb, 0, B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=3, dil=d, padL=d, mode=ops.EPI_DGRAD,
                                 r=dy, r_scale=math.sqrt(0.5), ymask=bits, ymask_rs=rs, drop_scale=1 / 0.95, a_split=pk.bwd_s,
                                 x_pair=True, tile_hint=hint)
        try:
            y = f()
        except RuntimeError as e:
            res["hint %d" % hint] = "n/a"
            continue
        var = L.dv3_debug_get(10)
        if ref is None:
            ref = y.clone()
        res["hint %d (%d)%s" % (hint, var, "" if torch.equal(y, ref) else " DIFFERS")] = "%.1f us" % graph_time(f)
    fl = 2.0 * B * T * C * 3 * 2 * C
    print("B=%d C=%d T=%d d=%d (floor at 833 TF: %.1f us):" % (B, C, T, d, fl / 833e12 * 1e6), "  ".join("%s %s" % kv for kv in res.items()), flush=True)
