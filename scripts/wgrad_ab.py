# coding: utf-8
"""wgrad at the north-star layer shape (B=64, M=512, Cin=256, T=1024, k=3, masked): one workgroup per tap
(dv3_debug_set(2, 1)) against one workgroup for all taps (dv3_debug_set(2, 3)); bit equality at equal slab counts,
then each with its own best slab count; also T=800 and the encoder shape (M=1024, Cin=512, T=150)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvoice3_pytorch_amd import ops, _lib
from scripts.planes_ab import timeit

dev = torch.device("cuda:0")
lib = _lib.lib()
ops.set_gemm_precision("f16x3")
for (B, M, C, T, d) in ((64, 512, 256, 1024, 1), (64, 512, 256, 804, 3), (64, 1024, 512, 150, 27), (64, 512, 256, 201, 9)):
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    g = torch.randn(B, M, T, device=dev)
    bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
    k = 3
    chunks = B * ((T + 31) // 32)
    tiles1 = ((M + 127) // 128) * ((C + 127) // 128)
    outs = {}
    row = []
    for variant, tiles, slots in ((1, tiles1 * k, 512), (3, tiles1, 256)):
        lib.dv3_debug_set(2, variant)
        for S in sorted({ops._ksplit_count(chunks, tiles, slots=slots), 21}):
            o = torch.empty((S, k, M, C), device=dev)
            f = lambda: ops.wgrad_gemm(g, x, B=B, M=M, Cin=C, T=T, Tin=T, J=k, dil=d, padL=d, n_slabs=S, xmask=bits, xmask_rs=rs,
                                       drop_scale=1 / 0.95, split_bf16=True, k_split=True, out=o)
            t = timeit(f, iters=30, settle=20)
            outs[(variant, S)] = o.sum(0)
            if S == 21:
                outs[(variant, "raw21")] = o.clone()
            row.append("v%d S=%d %.1f us (%d)" % (variant, S, t, lib.dv3_debug_get(11)))
    lib.dv3_debug_set(2, 0)
    same = torch.equal(outs[(1, "raw21")], outs[(3, "raw21")])
    print("B=%d M=%d C=%d T=%d d=%d | %s | bit-identical at S=21: %s" % (B, M, C, T, d, " | ".join(row), same))
