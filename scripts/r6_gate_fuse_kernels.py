# coding: utf-8
"""Round 6: what the fused gate backward costs and saves per launch (hipGraph-timed, scripts/r5_common.graph_time).
For a Conv1dGLU layer shape: the stand-alone gate backward; the input-gradient launch plain / reading pair words / with
the producer's gate backward in its tail / both; the weight-gradient launch on fp32 / pair-word g.
argv: B C T [dil]   (default: the north star 64 256 1024 and the converter's 64 256 804, 64 512 804)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import math
import torch
from scripts.r5_common import graph_time, dev, L
from deepvoice3_pytorch_amd import ops

ops.set_gemm_precision("f16x3")


def run(B, C, T, d=1, p=0.05):
    torch.manual_seed(0)
    k = 3
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True)
    x = torch.randn(B, C, T, device=dev)
    dy = torch.randn(B, C, T, device=dev)
    ab = torch.randn(B, 2 * C, T, device=dev)
    ops.dropout_state.manual_seed(3)
    bits, rs = ops.dropout_bits(B * C, T, p, dev)
    dab, _, part = ops.gate_bwd(dy, ab, None, B=B, C=C, T=T, mode=ops.EPI_GLU, residual=1)
    padL = d
    res = {}
    res["gate_bwd stand-alone"] = graph_time(lambda: ops.gate_bwd(dy, ab, None, B=B, C=C, T=T, mode=ops.EPI_GLU, residual=1))

    def dgrad(gate=False, xp=False, src=dab):
        gt = ops.GateFuse(ab, ops.EPI_GLU, 1, None, pair=True) if gate else None
        y = ops.conv_gemm(src, pk.bwd, pk.ldb, 0, B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=3, dil=d, padL=2 * d - padL,
                          mode=ops.EPI_DGRAD, r=dy, r_scale=math.sqrt(0.5), ymask=bits, ymask_rs=rs, drop_scale=1 / (1 - p),
                          a_split=pk.bwd_s, gate=gt, x_pair=xp)
        return y, gt
    y0, _ = dgrad()
    v0 = L.dv3_debug_get(10)
    y1, gt = dgrad(gate=True)
    v1 = L.dv3_debug_get(10)
    assert torch.equal(y0, y1)
    dabp = gt.dab           # pair words of the gate backward of (y1, ab)
    y2, _ = dgrad(xp=True, src=dabp)
    v2 = L.dv3_debug_get(10)
    res["dgrad plain (%d)" % v0] = graph_time(lambda: dgrad())
    res["dgrad + gate tail (%d)" % v1] = graph_time(lambda: dgrad(gate=True))
    res["dgrad pair-word input (%d)" % v2] = graph_time(lambda: dgrad(xp=True, src=dabp))
    res["dgrad pair-word input + gate tail"] = graph_time(lambda: dgrad(gate=True, xp=True, src=dabp))
    tiles = ((2 * C + 127) // 128) * ((C + 127) // 128)
    S = ops._ksplit_count(B * ((T + 31) // 32), tiles, slots=256)

    def wgrad(gp=False, src=dab):
        return ops.wgrad_gemm(src, x, B=B, M=2 * C, Cin=C, T=T, Tin=T, J=3, dil=d, padL=padL, n_slabs=S, xmask=bits,
                              xmask_rs=rs, drop_scale=1 / (1 - p), split_bf16=True, k_split=True, rows_of_slabs=True, g_pair=gp)
    wgrad()
    w0 = L.dv3_debug_get(11)
    res["wgrad fp32 g (%d)" % w0] = graph_time(lambda: wgrad())
    res["wgrad pair-word g"] = graph_time(lambda: wgrad(gp=True, src=dabp))
    print("B=%d C=%d T=%d d=%d:" % (B, C, T, d), "  ".join("%s %.1f us" % kv for kv in res.items()), flush=True)


if len(sys.argv) > 3:
    run(*[int(a) for a in sys.argv[1:5]])
else:
    for shape in ((64, 256, 1024), (64, 256, 804), (64, 512, 804), (64, 512, 150), (64, 256, 402), (16, 256, 804)):
        run(*shape)
