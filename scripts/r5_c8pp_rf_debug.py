# coding: utf-8
import math, torch
from r5_common import ops, L, dev
ops.set_gemm_precision("bf16"); ops.bf16_storage = True
B = 64
for (C, T, d, causal, k) in [(256, 1024, 1, False, 3), (96, 333, 9, True, 3)]:
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
    outs = []
    for rf in (0, 0, 1, 1):
        L.dv3_debug_set(30, rf)
        y8 = ops._c8_empty(B, C, T, dev).zero_(); ab = ops._c8_empty(B, 2 * C, T, dev).zero_(); ym8 = ops._c8_empty(B, C, T, dev).zero_(); dx8 = ops._c8_empty(B, C, T, dev).zero_()
        ekw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x8,
                   residual=1, a_split=pk.fwd_s, x_c8=x8, out_c8=True)
        mkw = dict(ekw, xmask_c8=keep8, drop_scale=1 / 0.95, ab=ab, y=ym8)
        dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=d, padL=(k - 1) * d - padL, mode=ops.EPI_DGRAD, r=x8, r_scale=0.7071,
                   drop_scale=1 / 0.95, a_split=pk.bwd_s, x_c8=gm8, out_c8=True, ymask_c8=keep8, y=dx8)
        ops.conv_gemm(None, None, pk.lda, pk.a_half, y=y8, **ekw)
        ops.conv_gemm(None, None, pk.lda, pk.a_half, **mkw)
        ops.conv_gemm(None, None, pk.ldb, 0, **dkw)
        torch.cuda.synchronize()
        outs.append([ops.from_c8(t_, n_) if hasattr(ops, "from_c8") else t_ for t_, n_ in ((y8, C), (ym8, C), (ab, 2 * C), (dx8, C))])
    L.dv3_debug_set(30, 0); L.dv3_debug_set(19, 128)
    names = ["y eval", "y masked", "ab", "dx"]
    for a_, b_, lab in ((0, 1, "rf0 vs rf0"), (2, 3, "rf1 vs rf1"), (0, 2, "rf0 vs rf1")):
        for i, nm in enumerate(names):
            ta, tb = outs[a_][i].float(), outs[b_][i].float()
            nd = int((ta != tb).sum())
            print("C=%d T=%d %s %-9s: %d differing of %d, max abs diff %.3e, nan %d/%d" % (C, T, lab, nm, nd, ta.numel(), float((ta - tb).abs().nan_to_num().max()), int(ta.isnan().sum()), int(tb.isnan().sum())))
