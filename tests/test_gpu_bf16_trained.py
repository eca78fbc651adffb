# coding: utf-8
"""-m gpu: the bf16 configurations (BASELINE configs 3 / 4: bf16 operands, bf16 channel-blocked storage) at TRAINED-LIKE
weights and at PRESET WIDTH.

End to end against the reference, the bf16 steps of the randomly initialised preset networks are only held to a sanity
bound (tests/test_gpu_preset_scale.py: 2e-1; measured 1e-3 .. 1.7e-1): a freshly initialised attention stack amplifies
any 2^-9 perturbation.  That says nothing about the weights a run actually spends its time at.  The fp32 CPU oracle is
too slow to train the preset-width model, but the `f16x3` path is pinned to the reference at preset size (outputs 2e-5,
train.train() steps), so it is a legitimate yardstick here:

  1. the preset-width deepvoice3_ljspeech model (reference train.py:685-759 step: losses, clip, Adam) is trained for
     WARM steps in `f16x3` on fixed synthetic batches with learnable structure, dropout on;
  2. at those weights the eval forward in `bf16` is compared with the eval forward in `f16x3`, output by output (sigmoid
     range [0, 1], normalised by the tensor maximum like every output tolerance of this repository):
       * rms error < RMS_TOL and the 99.99 % quantile < Q_TOL -- the size the per-layer half-ulp model predicts
         (~25 stacked roundings of 2^-9): measured 4e-4 .. 7e-4 rms, 5e-3 at the quantile
         (profiles/r06_bf16_trained_probe.txt);
       * the MAXIMUM is a different matter: a handful of elements of the converter output (7 in a million) sit where the
         trained network itself is ill-conditioned (|logit| up to 14, GLU gates in transition under activations of 25):
         the near-exact `bf16x3` arithmetic -- 1e-6-class perturbations -- shows the same elements 200-400 x above ITS rms.
         So the maximum is held to FWD_TOL, or to 3 x (max / rms of bf16x3 against f16x3) x (the bf16 rms; measured
         2.03 x): the
         network's own amplification at those weights, measured in the same test, times the typical bf16 error;
  3. both modes continue for CONT steps from the SAME state (weights, Adam moments, step count): the bf16 loss curve,
     in windows, must stay within BAND_MULT x the band two f16x3 continuations that differ only in their dropout draws
     span (the seed-to-seed noise at preset width), plus a small absolute floor.
The numbers are written to gpurun_out/bf16_trained.json (copied to profiles/).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WARM, CONT, WINDOW = 300, 200, 25
LR = 5e-4                    # the reference's peak learning rate (hparams.py:105), constant
B, TT, FRAMES, N_BATCH = 16, 100, 400, 4
FWD_TOL = 2e-2
RMS_TOL, Q_TOL = 1.5e-3, 2e-2
BAND_MULT, BAND_FLOOR = 2.0, 0.01


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _batches(hp):
    """fixed ragged batches whose targets are smooth functions of time and of the text (something to learn)"""
    import bench
    rng = np.random.RandomState(23)
    out = []
    for _ in range(N_BATCH):
        bt = bench.synth_batch(rng, B, TT, FRAMES, hp, fixed=True)
        tl = rng.randint(FRAMES // 2, FRAMES + 1, B) // 4 * 4
        il = rng.randint(TT // 2, TT + 1, B)
        tl[0], il[0] = FRAMES, TT
        L = bt["mel"].shape[1]
        t = torch.linspace(0, 1, L)[None, :, None]
        key = (bt["text"].float().mean(1) / hp["n_vocab"])[:, None, None]
        fm = torch.linspace(0, 1, hp["mel_dim"])[None, None, :]
        fl = torch.linspace(0, 1, hp["linear_dim"])[None, None, :]
        mel = 0.5 + 0.4 * torch.sin(9.0 * (t + key) + 4.0 * fm)
        y = 0.5 + 0.4 * torch.sin(9.0 * (t + key) + 4.0 * fl)
        done = torch.ones_like(bt["done"])
        for b in range(B):
            n, Li = int(tl[b]), int(il[b])
            mel[b, :1] = 0
            mel[b, 1 + n:] = 0
            y[b, :1] = 0
            y[b, 1 + n:] = 0
            done[b, :n // 4 - 1] = 0
            bt["text"][b, Li - 1] = 1
            bt["text"][b, Li:] = 0
            bt["text_positions"][b, Li:] = 0
        bt["mel"], bt["y"], bt["done"] = mel, y, done
        bt["input_lengths"], bt["target_lengths"] = il.astype(np.int64), tl.astype(np.int64)
        out.append(bt)
    return out


def _dev_batches(train_step, batches, dev):
    return [train_step.Batch.from_collate(bt["text"], bt["input_lengths"], bt["mel"], bt["y"], bt["text_positions"],
                                          bt["frame_positions"], bt["done"], bt["target_lengths"], None,
                                          downsample_step=4, device=dev) for bt in batches]


def _trainer(mode, hp, ga_sigma, sd, state, dev):
    """a Trainer in GEMM mode `mode` at the given weights and optimizer state"""
    from deepvoice3_pytorch_amd import builder, ops, train_step
    ops.set_gemm_precision(mode)
    model = builder.deepvoice3(**hp)
    model.load_state_dict(sd)
    model.to(dev)
    cfg = train_step.TrainConfig(max_positions=hp["max_positions"], guided_attention_sigma=ga_sigma, lr_schedule=None,
                                 initial_learning_rate=LR)
    tr = train_step.Trainer(model, cfg)
    if state is not None:
        tr.arena.exp_avg.copy_(state["m"])
        tr.arena.exp_avg_sq.copy_(state["v"])
        tr.adam_step, tr.global_step = state["adam_step"], state["global_step"]
    return model, tr


def _windows(x):
    return np.asarray(x, dtype=np.float64).reshape(-1, WINDOW).mean(1)


def test_bf16_at_trained_weights_forward_and_continued_training(dev):
    import bench
    from deepvoice3_pytorch_amd import ops, train_step
    bname, hp, ga_sigma = bench.PRESETS["deepvoice3_ljspeech"]
    hp = dict(hp)
    batches = _batches(hp)
    prev_mode = ops.gemm_precision()
    report = dict(model="deepvoice3_ljspeech preset width", B=B, text_len=TT, frames=FRAMES, lr=LR, warm_steps=WARM,
                  cont_steps=CONT, window=WINDOW)
    try:
        # ---- 1. warm-up in f16x3 ----
        from deepvoice3_pytorch_amd import builder
        torch.manual_seed(7)
        sd0 = {k: v.detach().clone() for k, v in builder.deepvoice3(**hp).state_dict().items()}
        model, tr = _trainer("f16x3", hp, ga_sigma, sd0, None, dev)
        dbs = _dev_batches(train_step, batches, dev)
        ops.dropout_state.manual_seed(41)
        warm = [tr.step(dbs[i % N_BATCH])["loss"] for i in range(WARM)]
        warm = torch.stack(warm).cpu().numpy().astype(np.float64)
        assert np.isfinite(warm).all()
        report["warm_loss_first_last"] = [round(float(warm[:10].mean()), 4), round(float(warm[-10:].mean()), 4)]
        assert warm[-10:].mean() < 0.6 * warm[:10].mean(), report["warm_loss_first_last"]       # it did train
        sd1 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        state = dict(m=tr.arena.exp_avg.clone(), v=tr.arena.exp_avg_sq.clone(), adam_step=tr.adam_step,
                     global_step=tr.global_step)
        tr.close()

        # ---- 2. eval forward at the trained weights: bf16 against f16x3 ----
        bt = batches[0]
        outs = {}
        for mode in ("f16x3", "bf16", "bf16x3"):
            model, tr = _trainer(mode, hp, ga_sigma, sd1, None, dev)
            model.eval()
            with torch.no_grad():
                mel_in = bt["mel"][:, 0::4, :].contiguous().to(dev)
                o = model(bt["text"].to(dev), mel_in, text_positions=bt["text_positions"].to(dev),
                          frame_positions=bt["frame_positions"].to(dev), input_lengths=bt["input_lengths"])
            outs[mode] = [t.float().cpu().numpy().astype(np.float64) for t in o]
            tr.close()

        def errs(a, b):
            e = np.abs(a - b) / max(np.abs(b).max(), 1e-30)
            return dict(max=float(e.max()), rms=float(np.sqrt((e ** 2).mean())), q9999=float(np.quantile(e, 0.9999)),
                        share_over_1e2=float((e > 1e-2).mean()))
        names = ("mel", "linear", "alignments", "done")
        fwd = {n: errs(a, b) for n, a, b in zip(names, outs["bf16"], outs["f16x3"])}
        cond = {n: errs(a, b) for n, a, b in zip(names, outs["bf16x3"], outs["f16x3"])}
        report["forward_bf16_vs_f16x3_at_trained_weights"] = {n: {k: float("%.3g" % v) for k, v in d.items()} for n, d in fwd.items()}
        report["forward_bf16x3_vs_f16x3_at_trained_weights"] = {n: {k: float("%.3g" % v) for k, v in d.items()} for n, d in cond.items()}
        # the same comparison at the INITIAL weights, for the record (the amplification the trained weights no longer have)
        outs0 = {}
        for mode in ("f16x3", "bf16"):
            model, tr = _trainer(mode, hp, ga_sigma, sd0, None, dev)
            model.eval()
            with torch.no_grad():
                o = model(bt["text"].to(dev), mel_in, text_positions=bt["text_positions"].to(dev),
                          frame_positions=bt["frame_positions"].to(dev), input_lengths=bt["input_lengths"])
            outs0[mode] = [t.float().cpu().numpy().astype(np.float64) for t in o]
            tr.close()
        report["forward_bf16_vs_f16x3_at_initial_weights"] = {
            n: float("%.3g" % (np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)))
            for n, a, b in zip(("mel", "linear", "alignments", "done"), outs0["bf16"], outs0["f16x3"])}

        # ---- 3. continue from the same state: two f16x3 dropout seeds span the band, bf16 must stay near it ----
        curves = {}
        for tag, mode, seed in (("f16x3_seed_a", "f16x3", 101), ("f16x3_seed_b", "f16x3", 202), ("bf16_seed_a", "bf16", 101)):
            model, tr = _trainer(mode, hp, ga_sigma, sd1, state, dev)
            dbs = _dev_batches(train_step, batches, dev)
            ops.dropout_state.manual_seed(seed)
            ls = [tr.step(dbs[i % N_BATCH])["loss"] for i in range(CONT)]
            curves[tag] = torch.stack(ls).cpu().numpy().astype(np.float64)
            tr.close()
        win = {k: _windows(v) for k, v in curves.items()}
        ref = win["f16x3_seed_a"]
        band = np.abs(win["f16x3_seed_b"] - ref) / ref
        devi = np.abs(win["bf16_seed_a"] - ref) / ref
        report["window_means"] = {k: [round(float(x), 5) for x in v] for k, v in win.items()}
        report["f16x3_seed_band_max_mean"] = [round(float(band.max()), 4), round(float(band.mean()), 4)]
        report["bf16_deviation_max_mean"] = [round(float(devi.max()), 4), round(float(devi.mean()), 4)]
        out_dir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out_dir):
            with open(os.path.join(out_dir, "bf16_trained.json"), "w") as f:
                json.dump(report, f)
        for k, v in curves.items():
            assert np.isfinite(v).all(), k
        for name in ("mel", "linear", "done"):
            f, c = fwd[name], cond[name]
            assert f["rms"] < RMS_TOL and f["q9999"] < Q_TOL, (name, report["forward_bf16_vs_f16x3_at_trained_weights"])
            amp = c["max"] / max(c["rms"], 1e-30)          # the network's own outlier amplification at these weights
            assert f["max"] < max(FWD_TOL, 3.0 * amp * f["rms"]), (name, f, c)
        assert devi.max() < BAND_MULT * band.max() + BAND_FLOOR, report
        assert devi.mean() < BAND_MULT * band.mean() + BAND_FLOOR, report
        assert win["bf16_seed_a"][-1] < 1.05 * win["bf16_seed_a"][0] + 1e-3       # still going down (or flat), not diverging
    finally:
        ops.set_gemm_precision(prev_mode)
