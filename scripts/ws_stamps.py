# coding: utf-8
"""Phase timestamps of one workgroup of the producer/consumer bf16x3 tap-GEMM (dv3_debug_set(3, 2) +
dv3_debug_set(1, 10)).  Developer tool, not a test."""
import math, sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvoice3_pytorch_amd import ops, _lib

dev = torch.device("cuda:0")
B, C, T, k = 64, 256, 1024, 3
dil = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
x = torch.randn(B, C, T, device=dev)
v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
bias = torch.zeros(2 * C, device=dev)
ops.set_gemm_precision("bf16x3")
pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
y = torch.empty(B, C, T, device=dev)


def launch():
    ops.conv_gemm(x, pk.fwd, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=dil,
                  padL=(k - 1) // 2 * dil, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x, residual=1, y=y,
                  tile_hint=29, a_split=pk.fwd_s)


_lib.call("dv3_debug_set", 3, 2)
for _ in range(200):      # warm clocks
    launch()
_lib.call("dv3_debug_set", 1, 10)
for _ in range(5):
    launch()
torch.cuda.synchronize()
_lib.call("dv3_debug_set", 1, 0)
_lib.call("dv3_debug_set", 3, 1)
buf = np.zeros((8, 192, 2), dtype=np.uint64)
_lib.call("dv3_debug_read", 1, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
nsteps = (C // 32) * k
S = 4
last = 1 + S * nsteps + 1
t = buf[:, :last + 1, 1].astype(np.int64)
r = buf[:, :last + 1, 0].astype(np.int64)
t0 = t[:, 0].min()
print("s_memtime ticks (x%.2f per 10 ns); slot-0 per wave (rel):" % ((t[0, last] - t[0, 0]) / float(r[0, last] - r[0, 0])), (t[:, 0] - t0).tolist())
d = [t[:, 2 + i:2 + i + S * nsteps:S] - t[:, 1 + i:1 + i + S * nsteps:S] for i in range(S)]
print("consumer wave 0: MFMAs  ", d[0][0].tolist())
print("consumer wave 0: barrier", d[3][0].tolist())
print("consumer wave 3: MFMAs  ", d[0][3].tolist())
print("consumer wave 3: barrier", d[3][3].tolist())
for w in (4, 7):
    for i, nm in enumerate(("storeA", "fetchA", "X items", "barrier")):
        print("producer wave %d %-8s" % (w, nm), d[i][w].tolist())
print("consumers mean: MFMAs %.0f barrier %.0f" % (d[0][:4].mean(), d[3][:4].mean()))
print("producers mean: storeA %.0f fetchA %.0f X %.0f barrier %.0f" % tuple(d[i][4:].mean() for i in range(4)))
print("main loop per wave:", (t[:, 1 + S * nsteps] - t[:, 0]).tolist(), " hand-off + epilogue:", (t[:, last] - t[:, 1 + S * nsteps]).tolist())
print("total:", int(t[:, last].max() - t0), " in 10-ns ticks:", int(r[:, last].max() - r[:, 0].min()))
