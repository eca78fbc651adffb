# coding: utf-8
"""Uninitialised-read check: one preset-size training forward+backward on clean memory and again with the caching
allocator's free pool filled with NaN; every output and gradient must be bit-identical (the step has no atomics).
Usage: python scripts/poison_check.py [preset ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvoice3_pytorch_amd import builder, ops, train_step
from tests.test_gpu_preset_scale import _preset, _batch, PRESET_NAMES

dev = torch.device("cuda:0")


def poison(gb=24):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    t = torch.full((gb << 28,), float("nan"), device=dev)
    torch.cuda.synchronize()
    del t


def run(preset, poisoned, hooks):
    bname, hp, sigma = _preset(preset)
    torch.manual_seed(12)
    model = getattr(builder, bname)(**hp).to(dev)
    bt, spk = _batch(hp)
    cfg = train_step.TrainConfig(max_positions=hp["max_positions"], guided_attention_sigma=sigma)
    trainer = train_step.Trainer(model, cfg)
    batch = train_step.Batch.from_collate(bt["text"], bt["input_lengths"], bt["mel"], bt["y"], bt["text_positions"],
                                          bt["frame_positions"], bt["done"], bt["target_lengths"], spk,
                                          downsample_step=4, device=dev)
    bad = []
    if hooks:
        def mk(name):
            def h(mod, inp, out):
                outs = out if isinstance(out, (tuple, list)) else (out,)
                for o in outs:
                    if torch.is_tensor(o) and o.is_floating_point() and not bool(torch.isfinite(o).all()):
                        bad.append(name)
            return h
        for n, m in model.named_modules():
            if n:
                m.register_forward_hook(mk(n))
    ops.dropout_state.manual_seed(777)
    trainer.arena.grad.zero_()
    if poisoned:
        poison()
    scal = trainer.forward_backward(batch)
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    scal = {k: float(v) for k, v in scal.items()}
    trainer.close()
    return scal, grads, bad


for preset in (sys.argv[1:] or PRESET_NAMES):
    s0, g0, _ = run(preset, False, False)
    s1, g1, bad = run(preset, True, True)
    diff = [(k, float((g0[k] - g1[k]).abs().max() / g0[k].abs().max().clamp_min(1e-30)) if bool(torch.isfinite(g1[k]).all()) else float("nan"))
            for k in g0 if not torch.equal(g0[k], g1[k])]
    print("== %s: loss clean %.7f poisoned %.7f | %d of %d gradients differ | first non-finite forward outputs: %s"
          % (preset, s0["loss"], s1["loss"], len(diff), len(g0), bad[:6]))
    for k, e in diff[:40]:
        print("   ", k, e)
