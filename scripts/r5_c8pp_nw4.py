# coding: utf-8
"""Round 5: conv_c8pp as two 4-wave workgroups per CU on 256 x 128 tiles (dv3_debug_set(34, 1); start delay of every other
block: (35, n x 64 cycles), (36, mask)) against the 8-wave 256 x 256 ping-pong form: bit-identity over layer forms,
graph-timed launches over the presets' shapes."""
import math
import torch
from r5_common import ops, L, dev, graph_time
from deepvoice3_pytorch_amd import modules

ops.set_gemm_precision("bf16")
ops.bf16_storage = True
L.dv3_debug_set(19, 1)
ok = True
for (kind, C, k, d, causal, T, B) in [("glu", 64, 3, 2, False, 75, 3), ("glu", 256, 3, 27, False, 150, 2), ("glu", 128, 3, 1, True, 100, 2),
                                      ("glu", 96, 3, 9, False, 61, 5), ("glu", 256, 3, 3, False, 800, 4), ("highway", 64, 3, 2, False, 75, 3),
                                      ("highway", 128, 1, 1, False, 50, 3), ("highway", 512, 3, 27, True, 150, 2), ("glu", 32, 3, 1, False, 33, 7),
                                      ("glu", 128, 1, 1, False, 50, 3), ("glu", 320, 3, 1, False, 130, 3), ("glu", 256, 3, 1, False, 1024, 8)]:
    torch.manual_seed(0)
    if kind == "highway":
        layer = modules.HighwayConv1d(C, C, k, dilation=d, causal=causal, dropout=0.1)
    else:
        layer = modules.Conv1dGLU(1, 16, C, C, k, dropout=0.2, dilation=d, causal=causal, residual=True)
    layer = layer.to(dev)
    with torch.no_grad():
        layer.conv.bias.uniform_(-0.1, 0.1)
    x = torch.randn(B, C, T, device=dev)
    res = {}
    for training in (False, True):
        layer.train(training)
        for nw4 in (0, 1):
            L.dv3_debug_set(34, nw4)
            L.dv3_debug_set(35, 4 if nw4 else 0)
            for p_ in layer.parameters():
                p_.grad = None
            xin = x.clone().requires_grad_(True)
            ops.dropout_state.manual_seed(5)
            y = ops.from_c8(layer(ops.to_c8(xin)))
            w = torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)
            (y * w).sum().backward()
            v = L.dv3_debug_get(10)
            res[(training, nw4)] = (y.detach(), xin.grad.detach(), [p_.grad.detach().clone() for p_ in layer.parameters()], v)
    L.dv3_debug_set(34, 0)
    same = True
    for training in (False, True):
        a, b = res[(training, 0)], res[(training, 1)]
        same &= torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(p0, p1) for p0, p1 in zip(a[2], b[2]))
    ok &= same
    print("%-8s C=%3d k=%d d=%2d causal=%d T=%4d B=%d variants %d / %d: %s" % (kind, C, k, d, causal, T, B, res[(True, 0)][3], res[(True, 1)][3],
          "BIT-EQUAL" if same else "DIFFERS (max %.3e)" % float((res[(True, 0)][0] - res[(True, 1)][0]).abs().max())), flush=True)
print("ALL BIT-EQUAL" if ok else "MISMATCH", flush=True)

B = 64
for (C, T, d, causal, k) in [(256, 1024, 1, False, 3), (256, 1024, 27, False, 3), (512, 150, 1, False, 3), (256, 200, 1, True, 3), (256, 400, 3, False, 3),
                             (256, 800, 1, False, 3), (512, 800, 3, False, 3), (256, 800, 1, False, 1)]:
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
    y8 = ops._c8_empty(B, C, T, dev); ab = ops._c8_empty(B, 2 * C, T, dev); ym8 = ops._c8_empty(B, C, T, dev); dx8 = ops._c8_empty(B, C, T, dev)
    ekw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x8,
               residual=1, a_split=pk.fwd_s, x_c8=x8, out_c8=True, y=y8)
    mkw = dict(ekw, xmask_c8=keep8, drop_scale=1 / 0.95, ab=ab, y=ym8)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=d, padL=(k - 1) * d - padL, mode=ops.EPI_DGRAD, r=x8, r_scale=0.7071,
               drop_scale=1 / 0.95, a_split=pk.bwd_s, x_c8=gm8, out_c8=True, ymask_c8=keep8, y=dx8)
    rows = []
    for (nw4, stag, mask) in ((0, 0, 1), (1, 0, 1), (1, 4, 1), (1, 16, 1), (1, 8, 8), (1, 8, 256), (0, 0, 1), (1, 4, 1)):
        L.dv3_debug_set(34, nw4); L.dv3_debug_set(35, stag); L.dv3_debug_set(36, mask)
        te = graph_time(lambda: ops.conv_gemm(None, None, pk.lda, pk.a_half, **ekw))
        tm = graph_time(lambda: ops.conv_gemm(None, None, pk.lda, pk.a_half, **mkw))
        td = graph_time(lambda: ops.conv_gemm(None, None, pk.ldb, 0, **dkw))
        rows.append("%s%s %.1f/%.1f/%.1f" % ("nw4" if nw4 else "nw8", (" s%d m%d" % (stag, mask)) if nw4 else "", te, tm, td))
    L.dv3_debug_set(34, 0); L.dv3_debug_set(35, 0); L.dv3_debug_set(36, 1)
    print("C=%3d T=%4d k=%d d=%2d (eval/train/dgrad us): %s" % (C, T, k, d, " | ".join(rows)), flush=True)
L.dv3_debug_set(19, 128)
