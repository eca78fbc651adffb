# coding: utf-8
"""one-wave-per-SIMD tap-GEMM (csrc/conv_gemm_w1.hip, tile_hint 30 / dv3_debug_set(12, 1)): bit-equality with the
shipped split kernels and timing at the north-star shape"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvoice3_pytorch_amd import ops, _lib
from scripts.planes_ab import timeit

dev = torch.device("cuda:0")
lib = _lib.lib()


def case(B, C, T, d, causal, masked, mode, train):
    ops.set_gemm_precision(mode)
    k = 3
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.randn(2 * C, device=dev) * 0.1
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=True)
    bits = rs = None
    if masked:
        ops.dropout_state.manual_seed(3)
        bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
    padL = (k - 1) * d if causal else d
    kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d, padL=padL, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x, residual=1,
              a_split=pk.fwd_s, xmask=bits, xmask_rs=rs or 0, drop_scale=1 / 0.95 if masked else 1.0)
    outs, abs_, times, var = [], [], [], []
    for hint in (0, 30):
        y = torch.empty(B, C, T, device=dev)
        ab = torch.empty(B, 2 * C, T, device=dev) if train else None
        ops.conv_gemm(x, None, pk.lda, pk.a_half, y=y, ab=ab, tile_hint=hint, **kw)
        var.append(lib.dv3_debug_get(10))
        outs.append(y); abs_.append(ab)
        if B * T >= 16384:
            times.append(timeit(lambda: ops.conv_gemm(x, None, pk.lda, pk.a_half, y=y, ab=ab, tile_hint=hint, **kw), iters=40, settle=30))
    same = torch.equal(outs[0], outs[1]) and (not train or torch.equal(abs_[0], abs_[1]))
    err = float((outs[0] - outs[1]).abs().max())
    # DGRAD form: gmat (B, 2C, T) -> dx (B, C, T) with the keep-mask on the output side and an addend
    gm = torch.randn(B, 2 * C, T, device=dev)
    dres = torch.randn(B, C, T, device=dev)
    dkw = dict(B=B, Cin=2 * C, Tin=T, M=C, Tout=T, J=k, dil=d, padL=(k - 1) * d - padL, mode=ops.EPI_DGRAD, r=dres,
               ymask=bits, ymask_rs=rs or 0, drop_scale=1 / 0.95 if masked else 1.0, a_split=pk.bwd_s)
    douts, dt = [], []
    for hint in (0, 30):
        dx = torch.empty(B, C, T, device=dev)
        ops.conv_gemm(gm, None, pk.ldb, 0, y=dx, tile_hint=hint, **dkw)
        douts.append(dx)
        if B * T >= 16384:
            dt.append(timeit(lambda: ops.conv_gemm(gm, None, pk.ldb, 0, y=dx, tile_hint=hint, **dkw), iters=40, settle=30))
    dsame = torch.equal(douts[0], douts[1])
    print("%-6s B=%d C=%d T=%d d=%d causal=%d masked=%d train=%d | fwd identical %s (max diff %.1e) variants %s %s | dgrad identical %s (%.1e) %s"
          % (mode, B, C, T, d, causal, masked, train, same, err, var, ["%.1f us" % t for t in times], dsame,
             float((douts[0] - douts[1]).abs().max()), ["%.1f us" % t for t in dt]), flush=True)


for mode in ("f16x3", "bf16x3"):
    case(3, 64, 75, 2, False, True, mode, True)
    case(2, 256, 150, 27, False, False, mode, False)
    case(2, 128, 100, 1, True, True, mode, True)
    case(64, 256, 1024, 1, False, False, mode, False)
    case(64, 256, 1024, 3, False, True, mode, True)
