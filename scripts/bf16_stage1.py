# coding: utf-8
"""bf16 storage, stage 1: Conv1dGLU forward at the north-star shape with bf16 x / residual / y / pre-gate save against
the fp32-storage bf16 kernel and the oracle"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvoice3_pytorch_amd import ops, _lib
from oracle import dv3_oracle as O
from scripts.planes_ab import timeit, x, v, g, bias, B, C, T, k, dev, lib
from tests.util import rel_err

ops.set_gemm_precision("bf16")
pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
xb = x.to(torch.bfloat16)
sd = {"l.conv.weight_v": v.cpu(), "l.conv.weight_g": g.cpu(), "l.conv.bias": bias.cpu()}
for d in (1, 27):
    for train in (False, True):
        kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=d, mode=ops.EPI_GLU, Cg=C, bias=bias, residual=1, a_split=pk.fwd_s)
        y32 = torch.empty(B, C, T, device=dev); ab32 = torch.empty(B, 2 * C, T, device=dev) if train else None
        y16 = torch.empty(B, C, T, device=dev, dtype=torch.bfloat16); ab16 = torch.empty(B, 2 * C, T, device=dev, dtype=torch.bfloat16) if train else None
        t32 = timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, r=x, y=y32, ab=ab32, **kw))
        t16 = timeit(lambda: ops.conv_gemm(xb, None, pk.lda, pk.a_half, r=xb, y=y16, ab=ab16, **kw))
        tmix = timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, r=x, y=y16, ab=ab16, **kw))
        O.set_operand_rounding("bf16")
        want = O.conv1d_glu(sd, "l", xb.float().cpu(), k, d, False, True)
        O.set_operand_rounding(None)
        e16 = rel_err(y16.float().cpu(), want)
        e32 = rel_err(y32.cpu(), O.conv1d_glu(sd, "l", x.cpu(), k, d, False, True))
        eab = rel_err(ab16.float().cpu(), ab32.cpu()) if train else 0.0
        print("d=%d train=%d | fp32 storage %.1f us (err %.1e) | bf16 storage %.1f us (err vs bf16-rounded oracle %.1e, ab vs fp32-storage %.1e) | fp32 in -> bf16 out %.1f us | variant %d"
              % (d, train, t32, e32, t16, e16, eab, tmix, lib.dv3_debug_get(10)))
