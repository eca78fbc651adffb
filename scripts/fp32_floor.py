# coding: utf-8
"""How far apart are two fp32-class evaluations of the preset-size network?  The CPU oracle in fp32 against
the same oracle in fp64 (ground truth): eval-forward outputs and, for one training step with a fixed dropout
mask, every parameter gradient (max-rel per tensor, as the GPU parity test measures it).  This is the yardstick
the preset-scale GPU tolerances are stated against.  Build container only (CPU)."""
import json, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dv3_oracle as O
from tests.util import rel_err
from tests.test_gpu_preset_scale import _batch, _preset
from deepvoice3_pytorch_amd import builder

out = {}
for preset in ("deepvoice3_ljspeech", "nyanko_ljspeech", "deepvoice3_vctk"):
    bname, hp, sigma = _preset(preset)
    torch.manual_seed(12)
    model = getattr(builder, bname)(**hp)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    spec = O.build_spec(bname, **hp)
    bt, spk = _batch(hp)
    mel_ds = bt["mel"][:, 0::4, :].contiguous()
    masks = {}
    g = torch.Generator().manual_seed(3)

    def drop(site, t, p, layout):
        if site not in masks:
            masks[site] = (torch.rand(t.shape, generator=g) >= p)
        return t * masks[site].to(t.dtype) / (1 - p)
    res = {}
    for dt in (torch.float32, torch.float64):
        sdc = {k: (v.to(dt) if v.dtype.is_floating_point else v).clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
        with torch.no_grad():
            ev = O.model_forward(sdc, spec, bt["text"], mel_ds.to(dt), spk, bt["text_positions"], bt["frame_positions"], bt["input_lengths"])
        o = O.model_forward(sdc, spec, bt["text"], mel_ds.to(dt), spk, bt["text_positions"], bt["frame_positions"], bt["input_lengths"], drop=drop)
        lhp = dict(outputs_per_step=1, downsample_step=4, masked_loss_weight=0.5, binary_divergence_weight=0.1, use_guided_attention=True, guided_attention_sigma=sigma)
        loss, parts = O.train_losses(spec, lhp, o, mel_ds.to(dt), bt["y"].to(dt), bt["done"].to(dt), bt["input_lengths"], bt["target_lengths"])
        loss.backward()
        res[dt] = (ev, {k: v.grad for k, v in sdc.items() if v.grad is not None}, float(loss))
    e32, g32, l32 = res[torch.float32]
    e64, g64, l64 = res[torch.float64]
    fw = {n: rel_err(a, b) for n, a, b in zip(("mel", "linear", "alignments", "done"), e32, e64)}
    ge = sorted(((rel_err(g32[k], g64[k]), k) for k in g64 if float(g64[k].abs().max()) > 0), reverse=True)
    l2 = sorted(((float((g32[k].double() - g64[k]).norm() / g64[k].norm()), k) for k in g64 if float(g64[k].norm()) > 0), reverse=True)
    out[preset] = dict(forward=fw, loss_rel=abs(l32 - l64) / abs(l64), grad_maxrel_top=ge[:5], grad_l2rel_top=l2[:5],
                       grad_maxrel_median=ge[len(ge) // 2][0])
    print(preset, json.dumps(out[preset], indent=1))
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_fp32_floor.json"), "w"), indent=1)
