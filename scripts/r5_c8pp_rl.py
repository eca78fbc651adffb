# coding: utf-8
"""Round 5: conv_c8pp with the residual read from LDS (dv3_debug_set(32, 1)): bit-identity with the residual read from
memory, over layer forms (GLU / highway, causal, masked, channel counts that do not fill a tile, 1 x 1), and graph-timed
launches over the presets' shapes."""
import math
import torch
from r5_common import ops, L, dev, graph_time
from deepvoice3_pytorch_amd import modules

ops.set_gemm_precision("bf16")
ops.bf16_storage = True
L.dv3_debug_set(19, 1)           # the 256 x 256 kernel wherever eligible
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
        for rl in (0, 1):
            L.dv3_debug_set(32, rl)
            for p_ in layer.parameters():
                p_.grad = None
            xin = x.clone().requires_grad_(True)
            ops.dropout_state.manual_seed(5)
            y = ops.from_c8(layer(ops.to_c8(xin)))
            v = L.dv3_debug_get(10)
            w = torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)
            (y * w).sum().backward()
            res[(training, rl)] = (y.detach(), xin.grad.detach(), [p_.grad.detach().clone() for p_ in layer.parameters()], v)
    L.dv3_debug_set(32, 0)
    same = True
    for training in (False, True):
        a, b = res[(training, 0)], res[(training, 1)]
        same &= torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(p0, p1) for p0, p1 in zip(a[2], b[2]))
    ok &= same
    print("%-8s C=%3d k=%d d=%2d causal=%d T=%4d B=%d variant %d: %s" % (kind, C, k, d, causal, T, B, res[(False, 1)][3],
          "BIT-EQUAL" if same else "DIFFERS (max %.3e)" % float((res[(True, 0)][0] - res[(True, 1)][0]).abs().max())), flush=True)
print("ALL BIT-EQUAL" if ok else "MISMATCH", flush=True)

B = 64
for (C, T, d, causal, k) in [(256, 1024, 1, False, 3), (256, 1024, 27, False, 3), (512, 150, 1, False, 3), (256, 400, 3, False, 3),
                             (256, 800, 1, False, 3), (512, 800, 3, False, 3), (256, 800, 1, False, 1)]:
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.randn(2 * C, device=dev) * 0.1
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False, split_only=True)
    x8 = ops.to_c8(x)
    ops.dropout_state.manual_seed(3)
    keep8 = ops.dropout_keep_c8(B, C, T, 0.05, dev)
    padL = (k - 1) * d if causal else (k - 1) // 2 * d
    y8 = ops._c8_empty(B, C, T, dev); ab = ops._c8_empty(B, 2 * C, T, dev); ym8 = ops._c8_empty(B, C, T, dev)
    ekw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x8,
               residual=1, a_split=pk.fwd_s, x_c8=x8, out_c8=True, y=y8)
    mkw = dict(ekw, xmask_c8=keep8, drop_scale=1 / 0.95, ab=ab, y=ym8)
    out = []
    for rnd in range(2):
        for rl in (0, 1):
            L.dv3_debug_set(32, rl)
            te = graph_time(lambda: ops.conv_gemm(None, None, pk.lda, pk.a_half, **ekw))
            tm = graph_time(lambda: ops.conv_gemm(None, None, pk.lda, pk.a_half, **mkw))
            out.append((rl, te, tm))
    L.dv3_debug_set(32, 0)
    fl = 2.0 * B * T * (2 * C) * (k * C)
    e0, e1 = min(o[1] for o in out if o[0] == 0), min(o[1] for o in out if o[0] == 1)
    m0, m1 = min(o[2] for o in out if o[0] == 0), min(o[2] for o in out if o[0] == 1)
    print("C=%3d T=%4d k=%d d=%2d: eval %6.1f -> %6.1f us (%.2f, %.0f TF = %.3f of 2.5 PF)   train fwd %6.1f -> %6.1f us (%.2f)" % (
        C, T, k, d, e0, e1, e1 / e0, fl / e1 / 1e6, fl / e1 / 1e6 / 2500.0, m0, m1, m1 / m0), flush=True)
L.dv3_debug_set(19, 128)
