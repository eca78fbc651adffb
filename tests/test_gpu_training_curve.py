# coding: utf-8
"""-m gpu: do the GEMM modes TRAIN alike?  (missing #4 / next #7.)

Per-launch pins prove that a bf16 kernel computes what it says; they do not prove that 4 M frames/s of bf16 steps
optimise like fp32 ones.  Here a small preset-like DeepVoice3 (dropout on, the reference's loss block and optimiser:
train.py:604-785) is trained for STEPS steps on fixed synthetic batches three ways from identical initial weights:

    f16x3   the default HIP path (BASELINE config 2's arithmetic)
    bf16    the HIP path of BASELINE configs 3/4 (bf16 operands, bf16 channel-blocked activation storage)
    oracle  oracle/dv3_oracle.py in fp32 on the CPU (torch's own dropout: the three runs draw different masks, so the
            comparison is statistical, as SURVEY.md section 7 says it must be)

and the loss trajectories, averaged over windows of WINDOW steps, must stay inside a stated band of the oracle's and
end within END_TOL of it.  The curves are written to gpurun_out/ (copied to profiles/r04_training_curves.json).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import dv3_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, WINDOW = 240, 20
# Two fp32 oracle runs that differ only in their dropout draws are themselves up to 6.6 % apart in single windows, 2.8 %
# on average and 2.2 % at the end on the GPU box (4.9 / 1.8 / 1.0 % on the build container: the trajectories are chaotic
# in the rounding of the host's thread count too; recorded as `oracle_seed_noise` in the report).  First measured HIP
# deviations (profiles/r04a_training_curves_first_run.json): f16x3 7.0 / 3.4 / 4.4 %, bf16 7.4 / 3.7 / 1.8 % -- i.e. at
# the noise.  The bands sit at about twice the noise: a mode that trains differently (a wrong gradient scale, a dead
# layer, a loss that stalls) misses them by far more.
BAND = 0.15        # every window mean within 15 % of the oracle's window mean
MEAN_DEV = 0.07    # window deviations averaged over the run within 7 %
END_TOL = 0.08     # mean of the last two windows within 8 %
LR = 1e-3          # constant (the Noam warm-up would keep lr below 6e-5 for the whole run and nothing would move)

HP = dict(n_vocab=40, embed_dim=64, mel_dim=32, linear_dim=65, r=1, downsample_step=4, n_speakers=1, padding_idx=0,
          dropout=0.05, kernel_size=3, encoder_channels=128, decoder_channels=64, converter_channels=64,
          use_memory_mask=True, force_monotonic_attention=True, use_decoder_state_for_postnet_input=True,
          max_positions=128, key_projection=True, value_projection=True)
LHP = dict(outputs_per_step=1, downsample_step=4, masked_loss_weight=0.5, binary_divergence_weight=0.1,
           use_guided_attention=True, guided_attention_sigma=0.2)
N_BATCH, B, TT, FRAMES = 4, 8, 20, 56


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _small_batches():
    """N_BATCH ragged batches at the test's shapes (text <= TT, frames <= FRAMES)"""
    import bench
    rng = np.random.RandomState(11)
    out = []
    for _ in range(N_BATCH):
        bt = bench.synth_batch(rng, B, TT, FRAMES, HP, fixed=True)
        tl = rng.randint(FRAMES // 2, FRAMES + 1, B) // 4 * 4
        il = rng.randint(TT // 2, TT + 1, B)
        tl[0], il[0] = FRAMES, TT
        L = bt["mel"].shape[1]
        t = torch.linspace(0, 1, L)[None, :, None]
        key = (bt["text"].float().mean(1) / HP["n_vocab"])[:, None, None]
        fm = torch.linspace(0, 1, HP["mel_dim"])[None, None, :]
        fl = torch.linspace(0, 1, HP["linear_dim"])[None, None, :]
        mel = 0.5 + 0.4 * torch.sin(6.0 * (t + key) + 4.0 * fm)
        y = 0.5 + 0.4 * torch.sin(6.0 * (t + key) + 4.0 * fl)
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


def _init_state():
    from deepvoice3_pytorch_amd import builder
    torch.manual_seed(3)
    return {k: v.detach().clone() for k, v in builder.deepvoice3(**HP).state_dict().items()}


def _run_hip(mode, sd0, batches, dev):
    from deepvoice3_pytorch_amd import builder, ops, train_step
    prev = ops.set_gemm_precision(mode)
    try:
        model = builder.deepvoice3(**HP)
        model.load_state_dict(sd0)
        model.to(dev)
        cfg = train_step.TrainConfig(max_positions=HP["max_positions"], lr_schedule=None, initial_learning_rate=LR)
        tr = train_step.Trainer(model, cfg)
        ops.dropout_state.manual_seed(5)
        dbs = [train_step.Batch.from_collate(bt["text"], bt["input_lengths"], bt["mel"], bt["y"], bt["text_positions"],
                                             bt["frame_positions"], bt["done"], bt["target_lengths"], None,
                                             downsample_step=4, device=dev) for bt in batches]
        losses = []
        for it in range(STEPS):
            losses.append(tr.step(dbs[it % len(dbs)])["loss"])
        out = torch.stack(losses).cpu().numpy().astype(np.float64)
        tr.close()
        return out
    finally:
        ops.set_gemm_precision(prev)


def _run_oracle(sd0, batches, seed=17):
    # a toy model on a many-core host: torch's default thread count (one per core: 256 on the GPU box) makes every small
    # op a thread-pool round trip -- 8 threads run these 240 steps in ~15 s instead of minutes
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(min(8, prev_threads))
    try:
        return _run_oracle_steps(sd0, batches, seed)
    finally:
        torch.set_num_threads(prev_threads)


def _run_oracle_steps(sd0, batches, seed):
    spec = O.build_spec("deepvoice3", **HP)
    sd = {k: v.clone() for k, v in sd0.items()}
    frozen = ("seq2seq.decoder.embed_query_positions.weight", "seq2seq.decoder.embed_keys_positions.weight")
    names = [k for k in sd if k not in frozen]
    for k in names:
        sd[k].requires_grad_(True)
    m = {k: torch.zeros_like(sd[k]) for k in names}
    v = {k: torch.zeros_like(sd[k]) for k in names}
    torch.manual_seed(seed)

    def drop(site, t, p, layout):
        return torch.nn.functional.dropout(t, p, True)

    mels = [bt["mel"][:, 0::4, :].contiguous() for bt in batches]
    losses = []
    for it in range(STEPS):
        bt, mel = batches[it % len(batches)], mels[it % len(batches)]
        for k in names:
            sd[k].grad = None
        out = O.model_forward(sd, spec, bt["text"], mel, None, bt["text_positions"], bt["frame_positions"],
                              bt["input_lengths"], drop=drop)
        loss, _ = O.train_losses(spec, LHP, out, mel, bt["y"], bt["done"], bt["input_lengths"], bt["target_lengths"])
        loss.backward()
        with torch.no_grad():
            O.clip_and_adam([sd[k] for k in names], [sd[k].grad for k in names], [m[k] for k in names],
                            [v[k] for k in names], it + 1, LR)
        losses.append(float(loss.detach()))
    return np.asarray(losses, dtype=np.float64)


def _windows(x):
    return x.reshape(-1, WINDOW).mean(1)


def test_loss_trajectories_of_bf16_f16x3_and_the_fp32_oracle_agree(dev):
    batches = _small_batches()
    sd0 = _init_state()
    curves = {"oracle_fp32": _run_oracle(sd0, batches), "oracle_fp32_other_masks": _run_oracle(sd0, batches, seed=117)}
    for mode in ("f16x3", "bf16"):
        curves[mode] = _run_hip(mode, sd0, batches, dev)
    win = {k: _windows(v) for k, v in curves.items()}
    ref = win["oracle_fp32"]
    report = dict(steps=STEPS, window=WINDOW, lr=LR, band=BAND, mean_dev=MEAN_DEV, end_tol=END_TOL,
                  model="deepvoice3 enc128/dec64/conv64, B=8 x 4 fixed batches, dropout 0.05, constant lr",
                  window_means={k: [round(float(x), 5) for x in v] for k, v in win.items()},
                  curves={k: [round(float(x), 5) for x in v] for k, v in curves.items()})
    worst = {}
    for mode in ("f16x3", "bf16", "oracle_fp32_other_masks"):
        dev_rel = np.abs(win[mode] - ref) / ref
        end_rel = abs(win[mode][-2:].mean() - ref[-2:].mean()) / ref[-2:].mean()
        worst[mode] = dict(max_window_dev=round(float(dev_rel.max()), 4), mean_window_dev=round(float(dev_rel.mean()), 4),
                           end_dev=round(float(end_rel), 4))
    report["oracle_seed_noise"] = worst.pop("oracle_fp32_other_masks")
    report["deviation_from_oracle"] = worst
    report["loss_drop"] = {k: round(float(v[0] / v[-1]), 3) for k, v in win.items()}
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "training_curves.json"), "w") as f:
            json.dump(report, f)
    # the run must actually have trained: the loss falls by a clear factor
    assert ref[0] / ref[-1] > 2.5, report["loss_drop"]
    for mode in ("f16x3", "bf16"):
        assert np.isfinite(curves[mode]).all()
        assert win[mode][0] / win[mode][-1] > 2.5, (mode, report["loss_drop"])      # the HIP run trains too (3.7-4.0 x measured)
        assert worst[mode]["max_window_dev"] < BAND, (mode, worst)
        assert worst[mode]["mean_window_dev"] < MEAN_DEV, (mode, worst)
        assert worst[mode]["end_dev"] < END_TOL, (mode, worst)
