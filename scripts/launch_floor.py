# coding: utf-8
"""Floor of a chain of dependent tiny launches on this GPU: a decode program of ONE entry (a 4 -> 4 linear layer, one
workgroup) issued for many steps by dv3_decode_program_launch, against the real 17-entry program."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepvoice3_pytorch_amd import ops, conv
from deepvoice3_pytorch_amd.decode_program import StepProgram
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
for Cin, Cout, B in ((4, 4, 1), (256, 256, 64), (256, 256, 4)):
    lin = conv.Linear(Cin, Cout).to(dev).eval()
    P = StepProgram(B, dev)
    x = P.buffer(B, Cin)
    P.conv_step(lin, x, ops.EPI_LINEAR, Cout)
    dones = P.buffer(4000, B, 1)
    for n in (200, 2000):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        P.decode_launched(x, torch.zeros(B, n, Cin, device=dev), dones, 0, 0)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("1-entry program %dx%d B=%d: %d steps, %.2f us per launch" % (Cin, Cout, B, n, dt / n * 1e6), flush=True)
