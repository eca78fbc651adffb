import os, sys
sys.path.insert(0, "/root/repo")
import torch
from scripts.r5_common import graph_time, dev, L
from deepvoice3_pytorch_amd import ops
ops.set_gemm_precision("bf16"); ops.bf16_storage = True
for B, C, M, T, J, d in ((64, 512, 1024, 804, 3, 3), (64, 256, 512, 804, 3, 1)):
    x8 = ops.to_c8(torch.randn(B, C, T, device=dev)); g8 = ops.to_c8(torch.randn(B, M, T, device=dev))
    keep = ops.dropout_keep_c8(B, C, T, 0.05, dev); keep = keep[0] if isinstance(keep, tuple) else keep
    tiles = ((M + 127) // 128) * ((C + 127) // 128)
    S = ops._ksplit_count(B * ((T + 31) // 32), tiles, slots=256, c8=True)
    for tr, name in ((1, "all"), (11, "no mfma"), (12, "no lds writes"), (13, "no global loads")):
        L.dv3_debug_set(52, tr)
        f = lambda: ops.wgrad_gemm_c8(g8, x8, B=B, M=M, Cin=C, T=T, J=J, dil=d, padL=d, n_slabs=S, xmask_c8=keep, drop_scale=1 / 0.95, rows_of_slabs=True)
        f()
        print(C, T, name, "%.1f us" % graph_time(f), flush=True)
L.dv3_debug_set(52, 1)
