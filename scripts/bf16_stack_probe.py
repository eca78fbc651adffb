"""Layer-by-layer distance between the HIP bf16 (c8) encoder and the oracle in bf16-operand + bf16-storage mode:
rms and max relative error and the fraction of stored values that differ, per layer of seq2seq.encoder."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from oracle import dv3_oracle as O
from deepvoice3_pytorch_amd import builder, ops
import bench
from tests.test_gpu_preset_scale import _batch, _preset

preset = sys.argv[1] if len(sys.argv) > 1 else "deepvoice3_ljspeech"
dev = torch.device("cuda:0")
bname, hp, _ = _preset(preset)
ops.set_gemm_precision("bf16")
torch.manual_seed(11)
model = getattr(builder, bname)(**hp).to(dev).eval()
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
spec = O.build_spec(bname, **hp)
bt, spk = _batch(hp)
se = F.embedding(spk, sd["embed_speakers.weight"]) if spk is not None else None
hip = []
enc = model.seq2seq.encoder
mods = enc.convolutions if hasattr(enc, "convolutions") else enc.convnet
for i, m in enumerate(mods):
    if isinstance(m, torch.nn.ReLU):
        continue
    def hook(mod, inp, out, i=i):
        C = getattr(mod, "out_channels", None) or mod.conv.out_channels // 2
        hip.append((i, (ops.from_c8(out, C) if ops.is_c8(out) else out).float().cpu()))
    m.register_forward_hook(hook)
ora = []
c1, cg, hw = O.conv1d, O.conv1d_glu, O.highway_conv1d
depth = [0]
def wrap(fn, name):
    def f(sd_, prefix, x, *a, **k):
        depth[0] += 1
        y = fn(sd_, prefix, x, *a, **k)
        depth[0] -= 1
        if depth[0] == 0:
            ora.append((prefix, y))
        return y
    return f
O.conv1d, O.conv1d_glu, O.highway_conv1d = wrap(c1, "c"), wrap(cg, "g"), wrap(hw, "h")
with torch.no_grad():
    enc(bt["text"].to(dev), lengths=bt["input_lengths"], speaker_embed=se.to(dev) if se is not None else None)
    O.set_operand_rounding("bf16", store=True)
    (O.dv3_encoder(sd, spec, bt["text"], se) if spec.kind == "deepvoice3" else O.ny_encoder(sd, spec, bt["text"]))
    O.set_operand_rounding(None)
print(len(hip), len(ora))
for (i, g), (pfx, w) in zip(hip, ora):
    if g.shape != w.shape:
        print(i, pfx, tuple(g.shape), tuple(w.shape)); continue
    # the oracle logs a conv before its ReLU / storage rounding: apply the same to compare stored values
    d = (g.double() - w.double())
    rms = float(d.norm() / w.double().norm())
    mx = float(d.abs().max() / w.abs().max())
    frac = float((d.abs() > 1e-3 * w.abs().clamp_min(1e-3 * float(w.abs().max()))).double().mean())
    print("%2d %-40s rms %.3e max %.3e differing %.3e" % (i, pfx, rms, mx, frac))
