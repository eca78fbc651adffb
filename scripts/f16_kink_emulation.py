# coding: utf-8
"""CPU experiment (no GPU): the nyanko preset train step in fp64, once exact and once with every conv / linear operand
rounded the way the f16x3 forward GEMMs round it (scaled fp16 hi + lo, include/dv3hip.h), straight-through gradient.
For 3 of 4 dropout draws single bias gradients move by 1e-3..1e-2 of their tensor's max although the loss agrees to
8 digits: one ReLU pre-activation / one |y_hat - y| of the L1 loss landing on the other side of zero.  This is why
tests/test_gpu_preset_scale.py differentiates the branch the HIP forward took (_KinkPins).
Output kept in profiles/r02_f16_kink_emulation.txt."""
import sys, math, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import dv3_oracle as O
import tests.test_gpu_preset_scale as TS
from deepvoice3_pytorch_amd import builder

preset = sys.argv[1] if len(sys.argv) > 1 else "nyanko_ljspeech"
bname, hp, sigma = TS._preset(preset)
torch.manual_seed(12)
model = getattr(builder, bname)(**hp)
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
spec = O.build_spec(bname, **hp)
bt, spk = TS._batch(hp)
mel_ds = bt["mel"][:, 0::4, :].contiguous()
lhp = dict(outputs_per_step=1, downsample_step=4, masked_loss_weight=0.5, binary_divergence_weight=0.1,
           use_guided_attention=True, guided_attention_sigma=sigma)
stats = {"max": 0.0, "clamped": 0}

def split16(t, shift):
    a = (t * (2.0 ** shift))
    stats["max"] = max(stats["max"], float(a.abs().max()) / 2.0 ** shift) if shift == 4 else stats["max"]
    if shift == 4:
        stats["clamped"] += int((a.abs() > 65504).sum())
    a = a.clamp(-65504, 65504)
    hi = a.to(torch.float16).to(t.dtype)
    lo = (a - hi).to(torch.float16).to(t.dtype)
    return (hi + lo) / (2.0 ** shift)

class RoundAct(torch.autograd.Function):     # straight-through: rounded forward value, exact gradient
    @staticmethod
    def forward(ctx, t, shift):
        return split16(t, shift)
    @staticmethod
    def backward(ctx, g):
        return g, None

def run(dt, seed, emul):
    gen = torch.Generator().manual_seed(seed)
    masks = {}
    def drop(site, t, p, layout):
        if site not in masks:
            masks[site] = (torch.rand(t.shape, generator=torch.Generator().manual_seed(hash((site, seed)) % (2**31))) >= p)
        return t * masks[site].to(t.dtype) / (1 - p)
    sdc = {k: (v.to(dt) if v.dtype.is_floating_point else v).clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    calls = [0]
    def _r(t):
        if not emul:
            return t
        calls[0] += 1
        # conv1d/linear call _r(x) then _r(w): odd calls = activation, even = weight
        return RoundAct.apply(t, 4 if calls[0] % 2 == 1 else 8)
    O._r = _r
    out = O.model_forward(sdc, spec, bt["text"], mel_ds.to(dt), spk, bt["text_positions"], bt["frame_positions"], bt["input_lengths"], drop=drop)
    loss, parts = O.train_losses(spec, lhp, out, mel_ds.to(dt), bt["y"].to(dt), bt["done"].to(dt), bt["input_lengths"], bt["target_lengths"])
    loss.backward()
    return float(loss), {k: v.grad for k, v in sdc.items() if v.grad is not None}

for seed in range(4):
    stats.update(max=0.0, clamped=0)
    l0, g0 = run(torch.float64, seed, False)
    l1, g1 = run(torch.float64, seed, True)
    errs = sorted(((float((g1[k] - g0[k]).abs().max() / g0[k].abs().max().clamp_min(1e-300)), k) for k in g0 if float(g0[k].abs().max()) > 1e-5 * max(float(v.abs().max()) for v in g0.values())), reverse=True)
    print("seed %d loss %.8f vs %.8f  max|act| %.1f clamped %d  worst grads:" % (seed, l0, l1, stats["max"], stats["clamped"]), [(k, "%.1e" % e) for e, k in errs[:4]], flush=True)
