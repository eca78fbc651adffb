# coding: utf-8
"""bench.py -- mel-frames/sec/node of the DeepVoice3 training step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full optimisation step (forward + losses + backward + [all-reduce] + clip + Adam,
train.py:604-785) of builder=deepvoice3 preset=deepvoice3_ljspeech on a synthetic LJSpeech-shaped
batch resident in HBM.  value = sum over ranks of un-padded target frames per step / wall time per
step (max over ranks).  GEMM arithmetic: --gemm bf16x3 (default: fp32 operands split into hi+lo bf16
on the matrix cores, fp32 accumulate, 1e-4 parity with the fp32 reference), f32 (exact fp32 MFMA) or
bf16.  Prints ONE JSON line on rank 0 with these extra objects:
  roofline      the dominant kernel (the tap-GEMM, Conv1dGLU forward at the north-star shape
                B=64 x 256ch x 1024T, k=3) timed with HIP events on its launch stream;
                bound "mfma": algorithmic FLOPs / time vs the MFMA roof of the mode (bf16x3:
                2500/3 TF; f32: 157.3 TF); hbm_frac is the same launch against the 8 TB/s HBM roof;
                traffic = HBM bytes per launch from the PMC passes recorded under profiles/
  roofline_exact_f32   the exact-fp32 kernel on the same launch, for the record
  cpu_baseline  the CPU oracle port of the same train step (oracle/dv3_oracle.py: the reference's
                own torch-CPU ops) on this host's cores, a bounded sample of the same workload
  host_buffers  the step rate if the boundary is handed pinned host tensors instead (never `value`)
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# presets/deepvoice3_ljspeech.json of the reference, as train.build_model() forwards it
# (train.py:812-840: key_position_rate / query_position_rate / embedding_weight_std are NOT
# forwarded, so the builder defaults apply).
DV3_LJ = dict(n_vocab=149, embed_dim=256, mel_dim=80, linear_dim=513, r=1, downsample_step=4,
              n_speakers=1, speaker_embed_dim=16, padding_idx=0, dropout=1 - 0.95, kernel_size=3,
              encoder_channels=512, decoder_channels=256, converter_channels=256, use_memory_mask=True,
              trainable_positional_encodings=False, force_monotonic_attention=True,
              use_decoder_state_for_postnet_input=True, max_positions=512,
              speaker_embedding_weight_std=0.01, freeze_embedding=False, window_ahead=3,
              window_backward=1, key_projection=True, value_projection=True)
# presets/nyanko_ljspeech.json and presets/deepvoice3_vctk.json as train.build_model() forwards them
NYANKO_LJ = dict(n_vocab=149, embed_dim=128, mel_dim=80, linear_dim=513, r=1, downsample_step=4, n_speakers=1,
                 speaker_embed_dim=16, padding_idx=0, dropout=1 - 0.95, kernel_size=3, encoder_channels=256,
                 decoder_channels=256, converter_channels=256, use_memory_mask=True,
                 trainable_positional_encodings=False, force_monotonic_attention=True,
                 use_decoder_state_for_postnet_input=True, max_positions=512, speaker_embedding_weight_std=0.01,
                 freeze_embedding=False, window_ahead=3, window_backward=1, key_projection=False,
                 value_projection=False)
DV3_VCTK = dict(n_vocab=149, embed_dim=256, mel_dim=80, linear_dim=513, r=1, downsample_step=4, n_speakers=108,
                speaker_embed_dim=16, padding_idx=0, dropout=1 - 0.95, kernel_size=3, encoder_channels=512,
                decoder_channels=256, converter_channels=256, use_memory_mask=True,
                trainable_positional_encodings=False, force_monotonic_attention=True,
                use_decoder_state_for_postnet_input=True, max_positions=1024, speaker_embedding_weight_std=0.05,
                freeze_embedding=False, window_ahead=3, window_backward=1, key_projection=True,
                value_projection=True)
PRESETS = {"deepvoice3_ljspeech": ("deepvoice3", DV3_LJ, 0.2), "nyanko_ljspeech": ("nyanko", NYANKO_LJ, 0.2),
           "deepvoice3_vctk": ("deepvoice3_multispeaker", DV3_VCTK, 0.4)}
PEAK_F32_MFMA_TF = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md: fp32 matrix peak (= vector peak)
PEAK_BF16_MFMA_TF = 2500.0  # same guide: dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0


def synth_batch(rng, B, Tt, n_frames, hp, fixed=True):
    """Synthetic LJSpeech-shaped items padded exactly as train.collate_fn does (train.py:293-360):
    target length rounded up to r and downsample_step, plus b_pad*downsample_step leading frames."""
    r, ds = hp["r"], hp["downsample_step"]
    if fixed:
        text_lens = np.full(B, Tt)
        frame_lens = np.full(B, n_frames)
    else:
        text_lens = np.clip(rng.normal(100, 30, B), 20, 187).astype(np.int64)
        frame_lens = np.clip(rng.normal(566, 180, B), 120, 870).astype(np.int64)
    max_in = int(text_lens.max())
    max_t = int(frame_lens.max())
    if max_t % r:
        max_t += r - max_t % r
    if max_t % ds:
        max_t += ds - max_t % ds
    b_pad = r
    max_t += b_pad * ds
    text = np.zeros((B, max_in), dtype=np.int64)
    tpos = np.zeros((B, max_in), dtype=np.int64)
    mel = np.zeros((B, max_t, hp["mel_dim"]), dtype=np.float32)
    y = np.zeros((B, max_t, hp["linear_dim"]), dtype=np.float32)
    Td = max_t // r // ds
    done = np.ones((B, Td, 1), dtype=np.float32)
    for b in range(B):
        L, n = int(text_lens[b]), int(frame_lens[b])
        text[b, :L - 1] = rng.randint(2, hp["n_vocab"], L - 1)
        text[b, L - 1] = 1
        tpos[b, :L] = np.arange(1, L + 1)
        mel[b, b_pad:b_pad + n] = rng.rand(n, hp["mel_dim"])
        y[b, b_pad:b_pad + n] = rng.rand(n, hp["linear_dim"])
        done[b, :n // r // ds - 1] = 0
    fpos = np.tile(np.arange(1, Td + 1, dtype=np.int64)[None], (B, 1))
    return dict(text=torch.from_numpy(text), input_lengths=text_lens, mel=torch.from_numpy(mel),
                y=torch.from_numpy(y), text_positions=torch.from_numpy(tpos),
                frame_positions=torch.from_numpy(fpos), done=torch.from_numpy(done),
                target_lengths=frame_lens)


def conv_roofline(dev, iters=100, tile_hint=0, dil=1, mode=None):
    """Conv1dGLU forward at the north-star shape, one tap-GEMM launch per iteration, timed with
    HIP events on the stream it is launched on (torch's current stream = the stream ops.* enqueue on).
    mode "bf16x3": the split-bf16 kernel (3 bf16 MFMAs per product block) -> peak = 2500/3 TF of
    fp32-equivalent work; mode "f32": the exact fp32-MFMA kernel -> peak 157.3 TF."""
    from deepvoice3_pytorch_amd import ops
    mode = mode or ops.gemm_precision()
    prev = ops.set_gemm_precision(mode)
    B, C, T, k, d = 64, 256, 1024, 3, dil
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.zeros(2 * C, device=dev)
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
    y = torch.empty(B, C, T, device=dev)

    def launch():
        ops.conv_gemm(x, pk.fwd, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d,
                      padL=(k - 1) // 2 * d, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x, residual=1, y=y,
                      tile_hint=tile_hint, a_split=pk.fwd_s)
    for _ in range(100):         # the clock governor needs ~20 ms of this load to settle (first launches run ~20 % slower)
        launch()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters):
        launch()
    e1.record(s)
    torch.cuda.synchronize()
    ops.set_gemm_precision(prev)
    us = e0.elapsed_time(e1) * 1e3 / iters
    flops = 2.0 * B * T * (2 * C) * (k * C)                     # 51.54 GFLOP
    byts = 4.0 * (B * C * T * 2 + 2 * C * C * k + 2 * C)         # x + y + weights + bias: 135.8 MB
    tf = flops / (us * 1e-6) / 1e12
    x3 = pk.fwd_s is not None and tile_hint in (0,) + tuple(range(21, 27))
    peak = PEAK_BF16_MFMA_TF / 3.0 if x3 else PEAK_F32_MFMA_TF
    out = dict(bound="mfma",
               kernel=("conv_gemm_bf16x3_kernel (Conv1dGLU fwd B=64 C=256 T=1024 k=3; 128x256 8-wave tile)" if x3 else
                       "conv_gemm_f32_stream_kernel<2,2,2> (Conv1dGLU fwd B=64 C=256 T=1024 k=3)"),
               achieved=round(tf, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(tf / peak, 4),
               traffic=None, us_per_launch=round(us, 2), alg_flops=flops, alg_bytes=byts,
               hbm_gbs=round(byts / (us * 1e-6) / 1e9, 1),
               hbm_frac=round(byts / (us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4))
    if x3 and tile_hint == 0 and dil == 1:
        # HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, calibrated as
        # MI355X_MICROARCH.md prescribes): collected offline with rocprofv3 (scripts/pmc_hbm.sh) --
        # a counter pass can not run inside this process -- and committed under profiles/.
        try:
            hb = json.load(open(os.path.join(ROOT, "profiles", "r01b_conv_gemm_bf16x3_hbm.json")))
            out["traffic"] = hb["hbm_bytes_per_launch"]
            out["traffic_source"] = "profiles/r01b_conv_gemm_bf16x3_hbm.json (%s)" % hb["source"]
        except (IOError, OSError, KeyError, ValueError):
            pass
    if x3:
        out["peak_note"] = ("algorithmic fp32 FLOPs against the dense bf16 MFMA peak (2500 TF) / 3: the split-bf16 "
                            "kernel issues 3 bf16 MFMAs per product block; executed MFMA rate = 3 x achieved")
        out["mfma_executed_tflops"] = round(3 * tf, 1)
        out["frac_of_bf16_mfma_peak"] = round(3 * tf / PEAK_BF16_MFMA_TF, 4)
        out["x_fp32_matrix_peak"] = round(tf / PEAK_F32_MFMA_TF, 3)
    return out


def cpu_baseline(B, Tt, n_frames, max_seconds=25.0):
    """The oracle port of the reference train step on the host cores (bounded sample)."""
    from oracle import dv3_oracle as O
    hp = dict(DV3_LJ)
    spec = O.build_spec("deepvoice3", **hp)
    from deepvoice3_pytorch_amd import builder
    torch.manual_seed(0)
    sd = {k: v.detach().clone() for k, v in builder.deepvoice3(**hp).state_dict().items()}
    frozen = ("seq2seq.decoder.embed_query_positions.weight", "seq2seq.decoder.embed_keys_positions.weight")
    names = [k for k in sd if k not in frozen]
    for k in names:
        sd[k].requires_grad_(True)
    m = {k: torch.zeros_like(sd[k]) for k in names}
    v = {k: torch.zeros_like(sd[k]) for k in names}
    rng = np.random.RandomState(1234)
    bt = synth_batch(rng, B, Tt, n_frames, hp)
    mel = bt["mel"][:, 0::4, :].contiguous()
    lhp = dict(outputs_per_step=1, downsample_step=4, masked_loss_weight=0.5, binary_divergence_weight=0.1,
               use_guided_attention=True, guided_attention_sigma=0.2)
    g = torch.Generator().manual_seed(0)

    def drop(site, t, p, layout):     # F.dropout stand-in with the same cost profile (bernoulli_ + mul)
        return torch.nn.functional.dropout(t, p, True)

    def one(it):
        for k in names:
            sd[k].grad = None
        out = O.model_forward(sd, spec, bt["text"], mel, None, bt["text_positions"], bt["frame_positions"],
                              bt["input_lengths"], drop=drop)
        loss, _ = O.train_losses(spec, lhp, out, mel, bt["y"], bt["done"], bt["input_lengths"], bt["target_lengths"])
        loss.backward()
        with torch.no_grad():
            O.clip_and_adam([sd[k] for k in names], [sd[k].grad for k in names], [m[k] for k in names],
                            [v[k] for k in names], it + 1, 5e-4)
    one(0)
    t0 = time.time()
    n = 0
    while n < 20 and (time.time() - t0) < max_seconds:
        one(n + 1)
        n += 1
    dt = (time.time() - t0) / n
    frames = float(bt["target_lengths"].sum())
    return dict(value=round(frames / dt, 1), unit="mel-frames/s", cores=torch.get_num_threads(), kind="port",
                sample="%d train steps of the same workload (B=%d, Tt=%d, %d frames/item) through "
                       "oracle/dv3_oracle.py on the host, %.2f s/step" % (n, B, Tt, n_frames, dt),
                host_cpus=os.cpu_count())


def synth_bench(dev, args):
    """BASELINE.json configs[4]: synthesis.py's path (synthesis.py:42-73) for `--batch` concurrent
    utterances: greedy autoregressive decode (Decoder.incremental_forward) + Converter + Griffin-Lim
    vocoder on the device.  SURVEY.md 8(d) cfg5: equal text length 100, min = max decoder steps = 200
    -> 201 steps = 804 frames = 9.33 s of audio each; RTF = wall / audio seconds."""
    from deepvoice3_pytorch_amd import builder, audio
    hp = dict(DV3_LJ)
    torch.manual_seed(0)
    model = builder.deepvoice3(**hp).to(dev).eval()
    model.make_generation_fast_()
    dec = model.seq2seq.decoder
    dec.min_decoder_steps = dec.max_decoder_steps = 200
    dec.use_step_graph = not args.no_graph       # replay one hipGraph per decoder step
    B, Tt = args.batch, 100
    rng = np.random.RandomState(0)
    text = torch.from_numpy(rng.randint(2, hp["n_vocab"], (B, Tt))).to(dev)
    tpos = torch.arange(1, Tt + 1).repeat(B, 1).to(dev)
    acfg = audio.AudioConfig(griffin_lim_iters=args.gl_iters)

    def run():
        t = [time.perf_counter()]
        with torch.no_grad():
            mel, lin, align, done = model(text, text_positions=tpos)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        wav = audio.inv_spectrogram_batch(lin, acfg)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        return lin, wav, t
    for _ in range(max(1, args.warmup // 3)):
        run()
    tm, tv = [], []
    for _ in range(max(1, args.steps // 10)):
        lin, wav, t = run()
        tm.append(t[1] - t[0]); tv.append(t[2] - t[1])
    assert torch.isfinite(wav).all() and lin.shape[1] == 804
    audio_s = B * wav.shape[1] / acfg.sample_rate
    wall = float(np.mean(tm) + np.mean(tv))
    out = dict(metric="real-time factor (synthesis: AR decode + converter + Griffin-Lim, %d concurrent utterances)" % B,
               value=round(wall / audio_s, 6), unit="wall seconds per audio second", n_gpus=1, steps=len(tm),
               warmup=max(1, args.warmup // 3), ms_per_step=round(wall * 1e3, 2), higher_is_better=False,
               scaling="weak", vs_baseline=None, dtype="f32", data="synthetic text ids, random-init weights",
               config=dict(workload="builder=deepvoice3 preset=deepvoice3_ljspeech synthesis, Tt=100, 201 decoder steps "
                                    "= 804 frames per utterance", utterances=B, griffin_lim_iters=args.gl_iters, step_hipgraph=bool(dec.use_step_graph),
                           audio_seconds=round(audio_s, 1), model_ms=round(float(np.mean(tm)) * 1e3, 1),
                           vocoder_ms=round(float(np.mean(tv)) * 1e3, 1),
                           rtf_model_only=round(float(np.mean(tm)) / audio_s, 6),
                           ms_per_decoder_step=round(float(np.mean(tm)) * 1e3 / 201, 3)))
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20,
                    help="untimed steps (the GPU clocks need ~1 s of this load to settle)")
    ap.add_argument("--batch", type=int, default=64,
                    help="per-GPU batch (north-star shape: 64; the preset's batch_size is 16)")
    ap.add_argument("--preset", default="deepvoice3_ljspeech", choices=sorted(PRESETS),
                    help="BASELINE.json configs[1] (default) / [2] nyanko_ljspeech / [3] deepvoice3_vctk")
    ap.add_argument("--gemm", default=None, choices=["bf16x3", "f32", "bf16"],
                    help="GEMM arithmetic (default: DV3_GEMM or bf16x3 = split-bf16 MFMA, fp32 accumulate)")
    ap.add_argument("--text-len", type=int, default=150)
    ap.add_argument("--frames", type=int, default=800)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a hipGraph replay")
    ap.add_argument("--graph", action="store_true", help="force the whole-step hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--mode", default="train", choices=["train", "conv", "conv-ab", "synth"])
    ap.add_argument("--gl-iters", type=int, default=60, help="Griffin-Lim iterations (synth mode)")
    args = ap.parse_args()

    from deepvoice3_pytorch_amd import builder, train_step, ops, dist as dv3dist
    if args.gemm:
        ops.set_gemm_precision(args.gemm)
    pg, rank, world, local_rank = dv3dist.init_from_env()
    assert world == args.gpus or world == 1 and args.gpus == 1, "launch with torchrun for --gpus > 1"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    if args.mode == "synth":
        synth_bench(dev, args)
        return
    if args.mode == "conv-ab":     # A/B of kernel variants / tiles at the north-star shape
        for hint in (0, 21, 22, 1, 2, 11, 12):
            for dil in (1, 27):
                rf = conv_roofline(dev, iters=10, tile_hint=hint, dil=dil, mode="bf16x3")
                print("tile_hint=%2d dil=%2d  %8.1f us  %6.1f TFLOP/s  frac %.3f" % (hint, dil, rf["us_per_launch"], rf["achieved"], rf["frac"]))
        return
    if args.mode == "conv":
        rf = conv_roofline(dev, iters=max(args.steps, 10))
        print(json.dumps(dict(metric="conv1dglu_fwd_tflops", value=rf["achieved"], unit="TFLOP/s", n_gpus=1,
                              steps=args.steps, warmup=5, ms_per_step=rf["us_per_launch"] / 1e3,
                              higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32",
                              data="synthetic", config=dict(workload="Conv1dGLU fwd B=64 C=256 T=1024 k=3"),
                              roofline=rf)))
        return

    bname, hp0, ga_sigma = PRESETS[args.preset]
    hp = dict(hp0)
    torch.manual_seed(0)            # identical initial weights on every rank
    model = getattr(builder, bname)(**hp).to(dev)
    cfg = train_step.TrainConfig(max_positions=hp["max_positions"], guided_attention_sigma=ga_sigma)
    trainer = train_step.Trainer(model, cfg, process_group=pg)
    rng = np.random.RandomState(1234 + rank)
    bt = synth_batch(rng, args.batch, args.text_len, args.frames, hp)
    spk = torch.from_numpy(rng.randint(0, hp["n_speakers"], args.batch)) if hp["n_speakers"] > 1 else None
    batch = train_step.Batch.from_collate(bt["text"], bt["input_lengths"], bt["mel"], bt["y"],
                                          bt["text_positions"], bt["frame_positions"], bt["done"],
                                          bt["target_lengths"], spk, downsample_step=4, device=dev)
    trainer.check_lengths(batch)
    # Launch mode.  The step is ~1.9k kernel launches; at the north-star batch (64) the GPU stays ahead of
    # the host, eager launches are GPU-bound (measured 18.25 ms/step eager vs 18.71 ms replayed) and the
    # RCCL bucket all-reduces can be issued from autograd hooks on a side stream.  A whole-step hipGraph
    # pays only when the step is launch-bound: small per-GPU batches on one GPU (B=16: 9.5 ms replayed).
    use_graph = (args.graph or args.batch < 32) and not args.no_graph and world == 1
    runner = None
    if use_graph:
        try:
            runner = train_step.GraphedTrainer(trainer, batch, warmup=max(1, min(args.warmup, 3)))
        except Exception as e:      # capture not possible (e.g. collective not capturable): say so, go eager
            if rank == 0:
                import traceback
                traceback.print_exc()
                print("hipGraph capture failed (%s); running eager" % type(e).__name__, file=sys.stderr)
            use_graph = False
            torch.cuda.synchronize()

    def do_step():
        return runner.step() if use_graph else trainer.step(batch)

    for _ in range(args.warmup):
        scal = do_step()
    if pg is not None:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scal = do_step()
    torch.cuda.synchronize()
    if pg is not None:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    frames = torch.tensor([float(batch.n_frames)], dtype=torch.float64, device=dev)
    if pg is not None:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(frames, op=torch.distributed.ReduceOp.SUM)
    dt = float(tmax.item())
    loss = float(scal["loss"])
    if not math.isfinite(loss):
        raise RuntimeError("non-finite training loss (%r): the measurement is void" % loss)
    if rank != 0:
        return
    ms = dt / args.steps * 1e3
    value = float(frames.item()) / (dt / args.steps)
    mode_desc = {"bf16x3": "f32 (operands split hi+lo bf16 on the matrix cores, 3 MFMAs per product, fp32 accumulate; "
                           "1e-4 rel parity with the fp32 reference)",
                 "f32": "f32 (exact fp32 MFMA)",
                 "bf16": "bf16 (operands rounded to bf16 at the matrix cores, fp32 accumulate, fp32 master weights)"}
    out = dict(metric="mel-frames/sec/node (train step, %s)" % args.preset, value=round(value, 1),
               unit="mel-frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=round(ms, 3), higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype={"bf16x3": "f32", "f32": "f32", "bf16": "bf16"}[ops.gemm_precision()],
               dtype_note=mode_desc[ops.gemm_precision()],
               data="synthetic (fixed-shape LJSpeech-like: Tt=%d, %d frames/item; random-init weights)"
               % (args.text_len, args.frames),
               config=dict(workload="builder=%s preset=%s train step "
                                    "(fwd+losses+bwd+clip+Adam), synthetic LJSpeech-shaped batches" % (bname, args.preset),
                           per_gpu_batch=args.batch, global_batch=args.batch * world, text_len=args.text_len,
                           frames_per_item=args.frames, parallelism="dp%d" % world,
                           hipgraph=bool(use_graph), gemm=ops.gemm_precision(), final_loss=round(loss, 5)))
    # the boundary can also be handed host buffers (train.py:655-663 copies 8 tensors per step): time the
    # H2D of one pinned batch and report the rate with that copy serialised in front of every step
    pinned = {k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in bt.items()}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        train_step.Batch.from_collate(pinned["text"], pinned["input_lengths"], pinned["mel"], pinned["y"],
                                      pinned["text_positions"], pinned["frame_positions"], pinned["done"],
                                      pinned["target_lengths"], spk, downsample_step=4, device=dev)
        torch.cuda.synchronize()
    h2d_ms = (time.perf_counter() - t0) / 3 * 1e3
    out["host_buffers"] = dict(h2d_ms_per_batch=round(h2d_ms, 3),
                               value_with_serial_h2d=round(float(frames.item()) / ((ms + h2d_ms) * 1e-3), 1),
                               note="pinned host batch -> HBM copied synchronously before each step; never `value`")
    if not args.no_roofline:
        out["roofline"] = conv_roofline(dev, mode="bf16x3" if ops.gemm_precision() == "bf16" else None)
        if ops.gemm_precision() != "f32":      # the exact-fp32 kernel beside it, for the record
            rf = conv_roofline(dev, mode="f32")
            out["roofline_exact_f32"] = dict(kernel=rf["kernel"], achieved=rf["achieved"], peak=rf["peak"],
                                             frac=rf["frac"], us_per_launch=rf["us_per_launch"])
    if not args.no_cpu_baseline and world == 1 and args.preset == "deepvoice3_ljspeech":
        out["cpu_baseline"] = cpu_baseline(args.batch, args.text_len, args.frames)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
