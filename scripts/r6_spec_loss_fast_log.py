# coding: utf-8
"""Round 6: the tiled spectrogram loss (time-fastest prediction, bin-fastest target) with its three logarithms by
v_log_f32 (dv3_debug_set(57, 1), default) against logf: time (graph-timed), the four loss sums (relative), the gradient
(bit for bit: it takes no logarithm), and both against a double-precision evaluation of train.py:537-569."""
import sys

import torch

from r5_common import dev, graph_time, L
from deepvoice3_pytorch_amd import ops

for B, T, D, r in ((64, 804, 513, 4), (64, 804, 80, 4), (16, 804, 513, 4), (16, 804, 80, 4)):
    torch.manual_seed(D)
    yh = torch.sigmoid(torch.randn(B, D, T, device=dev) * 2).transpose(1, 2)      # (B, T, D) view of a BCT tensor
    y = torch.rand(B, T, D, device=dev)
    lengths = torch.randint(T // 2, T + 1, (B,), device=dev, dtype=torch.int32)
    res, us = {}, {}
    for sw in (0, 1):
        L.dv3_debug_set(57, sw)
        o4, g = ops.spec_loss_with_grad(yh, y, lengths, r=r)
        res[sw] = (o4.clone(), g.clone())
        us[sw] = graph_time(lambda: ops.spec_loss_with_grad(yh, y, lengths, r=r))
    L.dv3_debug_set(57, 1)
    rel = float(((res[0][0] - res[1][0]).abs() / res[0][0].abs()).max())
    same = torch.equal(res[0][1], res[1][1])
    # double-precision value of the binary divergence mean (train.py:537-556) on y_hat[:, :-r] / y[:, r:]
    a, t = yh[:, :-r].double(), y[:, r:].double()
    eps = 1e-8
    lg = torch.log(a + eps) - torch.log(1 - a + eps)
    z = (-t * lg + torch.log1p(torch.exp(lg))).mean()
    l1 = (a - t).abs().mean()
    print("B %3d T %4d D %4d | logf %7.1f us  v_log %7.1f us  %.3f | sums rel %.1e, gradient %s | fast out4 %s | double: l1 %.7f bd %.7f"
          % (B, T, D, us[0], us[1], us[1] / us[0], rel, "bit-identical" if same else "DIFFERS",
             [round(float(v), 7) for v in res[1][0]], float(l1), float(z)))
