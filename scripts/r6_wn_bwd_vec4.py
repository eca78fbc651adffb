# coding: utf-8
"""Round 6: dv3_weight_norm_bwd_f32 with the 16-byte slab gather against the 4-byte one (dv3_debug_set(51, v)), the step's
layer shapes, stand-alone from a hipGraph of 20 launches (scripts/r5_common.graph_time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
from deepvoice3_pytorch_amd import ops, _lib
from r5_common import graph_time
dev = torch.device("cuda:0")
L = _lib.lib()
print("%5s %5s %2s %3s | 4-byte us  16-byte us" % ("O", "I", "J", "S"))
for O_, I, J, S in [(1024, 512, 3, 22), (1024, 512, 3, 14), (512, 256, 3, 32), (512, 256, 3, 14), (512, 256, 3, 28), (256, 256, 1, 14),
                    (512, 512, 1, 16), (513, 512, 1, 8), (512, 128, 5, 16)]:
    slabs = torch.randn(J, O_, S, I, device=dev)
    v = torch.randn(O_, I, J, device=dev) * 0.1
    g = torch.rand(O_, 1, 1, device=dev) + 0.5
    scale = 1.0 / v.reshape(O_, -1).norm(dim=1)
    part = torch.randn(O_, 400, device=dev)
    into = (torch.zeros_like(v), torch.zeros_like(g), torch.zeros(O_, device=dev))
    t = []
    for sw in (0, 1):
        L.dv3_debug_set(51, sw)
        t.append(graph_time(lambda: ops.weight_norm_bwd(slabs, S, I, v, g, scale, part, 400, O_, I, J, rows_of_slabs=True, into=into, part_t=True),
                            per_graph=20, replays=4))
    L.dv3_debug_set(51, 1)
    print("%5d %5d %2d %3d | %8.1f  %8.1f" % (O_, I, J, S, t[0], t[1]), flush=True)
