# coding: utf-8
"""Phase timestamps of the 256 x 256 k16 ping-pong tap-GEMM (dv3_debug_set(13, 4)): real shader clock and cycles per
phase segment, one workgroup, north-star shape."""
import ctypes
import math
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepvoice3_pytorch_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
ops.set_gemm_precision("f16x3")
B, C, T, k = 64, 256, 1024, 3
torch.manual_seed(0)
x = torch.randn(B, C, T, device=dev)
v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
bias = torch.zeros(2 * C, device=dev)
pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
y = torch.empty(B, C, T, device=dev)
kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x,
          residual=1, a_split=pk.fwd_s, y=y, tile_hint=30)
L.dv3_debug_set(13, 4)
for _ in range(200):
    ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw)
torch.cuda.synchronize()
NS = 320
buf = (ctypes.c_ulonglong * (8 * NS * 2))()
_lib.call("dv3_debug_read", 2, buf, ctypes.sizeof(buf))
L.dv3_debug_set(13, 0)
a = np.frombuffer(buf, dtype=np.uint64).reshape(8, NS, 2).astype(np.int64)
nph = 48
last = 2 + 5 * nph + 1
for w in (0, 4):
    rt = a[w, last, 0] - a[w, 0, 0]
    cy = a[w, last, 1] - a[w, 0, 1]
    print("wave %d: kernel body %d cycles in %.2f us -> shader clock %.3f GHz" % (w, cy, rt / 100.0, cy / (rt * 10.0)))
names = ["LOAD: staging issued", "LOAD: fragment reads landed", "barrier after LOAD", "COMPUTE: 24 MFMAs issued", "barrier after COMPUTE"]
cyc = a[:, :, 1]
seg = np.zeros((8, nph, 5))
for ph in range(nph):
    b = 2 + 5 * ph
    for i in range(5):
        nxt = b + i + 1
        seg[:, ph, i] = cyc[:, nxt] - cyc[:, b + i]
print("prologue %.0f cycles, main loop %.0f, tail %.0f" % ((cyc[:, 1] - cyc[:, 0]).mean(), (cyc[:, 2 + 5 * nph] - cyc[:, 1]).mean(),
                                                            (cyc[:, last] - cyc[:, 2 + 5 * nph]).mean()))
for i, n in enumerate(names):
    print("%-32s mean %7.0f   early %7.0f  late %7.0f   (phases that convert an activation item: %7.0f, others %7.0f)" % (
        n, seg[:, 2:, i].mean(), seg[:4, 2:, i].mean(), seg[4:, 2:, i].mean(),
        seg[:, 2:, i][:, [p_ % 2 == 0 for p_ in range(2, nph)]].mean(),
        seg[:, 2:, i][:, [p_ % 2 == 1 for p_ in range(2, nph)]].mean()))
print("per phase total: %.0f cycles (both halves: interval = half of it)" % seg[:, 2:, :].sum(axis=2).mean())
