# coding: utf-8
"""bench.py -- mel-frames/sec/node of the DeepVoice3 training step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: spawns its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --dry-launch      rendezvous + bucketed all-reduce only (gloo on a CPU-only box)

One "step" = one full optimisation step (forward + losses + backward + [all-reduce] + clip + Adam,
train.py:604-785) of builder=deepvoice3 preset=deepvoice3_ljspeech (BASELINE.json configs[1]) on a synthetic
LJSpeech-shaped batch resident in HBM.  value = sum over ranks of un-padded target frames per step / wall
time per step (max over ranks).  GEMM arithmetic --gemm: f16x3 (default: fp32 operands split into scaled fp16
hi+lo for the forward GEMMs and bf16 hi+lo for the gradient GEMMs, three 16-bit MFMAs per product, fp32
accumulate; 1e-4 parity with the fp32 reference at the preset sizes), bf16x3, f32 (exact fp32 MFMA), bf16.
`dtype` names that arithmetic.  Rank 0 prints ONE JSON line; beside the contract's fields it carries

  roofline            the dominant kernel (forward tap-GEMM, Conv1dGLU at the north-star shape B=64 x 256ch x
                      1024T, k=3) timed with HIP events on its launch stream; bound "mfma": algorithmic FLOPs /
                      time vs the MFMA roof of the mode (three 16-bit MFMAs per product: 2500/3 TF; f32: 157.3
                      TF); traffic = HBM bytes per launch from the PMC passes under profiles/ (null until the
                      pass for THIS kernel has been collected)
  roofline_wgrad      the weight-gradient GEMM of the same layer, same way
  roofline_exact_f32  the exact-fp32 forward kernel on the same launch, for the record
  roofline_bf16_c8    the same layer as the bf16 configs run it (single-term bf16 MFMA, channel-blocked bf16
                      input / residual / output) against the 2500 TF bf16 roof and the 8 TB/s HBM roof
  step_flop_frac      whole step: algorithmic FLOPs per mel-frame x frames / step time vs the
                      same MFMA roof
  value_exact_f32     the same step with every GEMM on the exact fp32 MFMA chain (5 steps)
  launch_mode /       how the timed step was issued: "eager_two_streams", or "graph" with graph_form "segments" (a chain
  graph_form          of segment hipGraphs, the weight-gradient branch replayed on a real second stream) or "single";
                      probed per configuration (both timed, the faster kept; ties go to eager)
  configs             BASELINE.json configs[2..4] from the same process: nyanko_ljspeech bf16, deepvoice3_vctk bf16
                      (n_gpus = this run's), synthesis RTF (64 utterances); plus dv3lj_b16 (the preset's own batch),
                      dv3lj_b64_ragged / dv3lj_b16_ragged (LJSpeech-like length spread instead of full-length rows)
                      and ddp_world1 (the headline step with a one-rank RCCL group armed: what the bucketed
                      all-reduce path costs on one GPU, eager and replayed)
  input_pipeline      the same step fed by data.Prefetcher (rank-sharded length-bucketed sampler, pinned
                      staging, H2D + device-side collate on a side stream) instead of a resident batch
  host_buffers        the step rate if the boundary is handed pinned host tensors synchronously (never `value`)
  cpu_baseline        the reference's own train.train() on the host cores when the reference tree is present
                      (kind "reference"), else the CPU oracle port of the same step (kind "port"); bounded sample
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
# algorithmic forward+backward FLOPs per un-padded mel-frame at the bench shapes (FlopCounterMode)
MFLOP_PER_FRAME = {"deepvoice3_ljspeech": 57.1, "nyanko_ljspeech": 64.8, "deepvoice3_vctk": 52.9}
MFMA_RANDOM_OPERANDS_TF = 1681.0      # measured, profiles/r03_mfma_power.txt (zero operands: 2463)
PEAK_F32_MFMA_TF = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md: fp32 matrix peak (= vector peak)
PEAK_16BIT_MFMA_TF = 2500.0  # same guide: dense bf16 / fp16 MFMA peak
PEAK_HBM_GBS = 8000.0

DTYPE_NOTE = {
    "f16x3": "fp32 operands as scaled fp16 hi+lo (forward GEMMs) / bf16 hi+lo (gradient GEMMs) on the matrix cores, "
             "3 MFMAs per product, fp32 accumulate, fp32 storage; 1e-4 rel parity with the fp32 reference at preset sizes",
    "bf16x3": "fp32 operands as bf16 hi+lo on the matrix cores, 3 MFMAs per product, fp32 accumulate, fp32 storage",
    "f32": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32)",
    "bf16": "operands rounded to bf16 at the matrix cores, 1 MFMA per product, fp32 accumulate, fp32 master weights"}
BF16_STORAGE_NOTE = ("; activations, saved pre-gates and activation gradients of the conv stacks stored in HBM as "
                     "channel-blocked bf16 (c8, include/dv3hip.h)")


def dtype_note(mode):
    """the arithmetic / storage description that goes with a `dtype` token"""
    note = DTYPE_NOTE[mode]
    if mode == "bf16":
        from deepvoice3_pytorch_amd import ops
        note += BF16_STORAGE_NOTE if ops.bf16_storage else "; fp32 activations in HBM (DV3_BF16_STORAGE=0)"
    return note


def mfma_peak_tf(mode):
    return {"f32": PEAK_F32_MFMA_TF, "bf16": PEAK_16BIT_MFMA_TF}.get(mode, PEAK_16BIT_MFMA_TF / 3.0)


def synth_batch(rng, B, Tt, n_frames, hp, fixed=True, lengths=None):
    """Synthetic LJSpeech-shaped items padded exactly as train.collate_fn does (train.py:293-360):
    target length rounded up to r and downsample_step, plus b_pad*downsample_step leading frames.
    lengths: (text_lens, frame_lens) of the B items (a batch drawn by a sampler)"""
    r, ds = hp["r"], hp["downsample_step"]
    if lengths is not None:
        text_lens, frame_lens = np.asarray(lengths[0], dtype=np.int64), np.asarray(lengths[1], dtype=np.int64)
    elif fixed:
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


# -------------------------------------------------------------------------------------------------
# kernel rooflines (HIP events on the launch stream)
# -------------------------------------------------------------------------------------------------
_CAP_STREAM = {}


def _capture_stream():
    """ONE capture stream per device for every roofline capture of this process (each torch.cuda.Stream() would pin a
    67 MB stream-K workspace of its own in ops._sk_ws)"""
    d = torch.cuda.current_device()
    if d not in _CAP_STREAM:
        _CAP_STREAM[d] = torch.cuda.Stream()
    return _CAP_STREAM[d]


def _time_launches(launch, iters, settle=100, per_graph=25):
    """us per launch, HIP events on the stream the launches run on.  Round 5: the launches are captured into one hipGraph
    (`per_graph` of them) and the REPLAYS are timed -- eager launches through ctypes cost the host 30-120 us each
    (box dependent), so any kernel shorter than that measured as the host's issue rate (the 60 us bf16 tap-GEMM read
    63.8 us eager against 59.3 us replayed on one box, profiles/r05_ship_check.txt; a 30 us launch reads 123 us on a
    slow host).  Falls back to eager launches if the capture fails."""
    for _ in range(settle):      # the clock governor needs ~20 ms of this load to settle (first launches run ~20 % slower)
        launch()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream()      # the stream ops.* enqueue on
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        if os.environ.get("DV3_BENCH_EAGER_TIMING", "") == "1":      # counter passes (scripts/pmc_r5.sh): one launch per dispatch
            raise RuntimeError("eager timing requested")
        cap = _capture_stream()
        cap.wait_stream(s)
        #  the launches captured here must be the variants the training step runs: a capture on a stream that
        # has no stream-K workspace gets the tile-per-workgroup form (ops._streamk_ws never allocates inside a capture)
        from deepvoice3_pytorch_amd import ops as _ops
        _ops.prepare_streamk_ws(s.device, cap)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            for _ in range(per_graph):
                launch()
        torch.cuda.synchronize()
        replays = max(2, -(-iters // per_graph))
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        e0.record(s)
        for _ in range(replays):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
        _time_launches.last_mode = "hipGraph of %d launches, %d replays" % (per_graph, replays)
        return e0.elapsed_time(e1) * 1e3 / (replays * per_graph)
    except Exception as e:      # noqa: a capture that fails must not void the line
        if os.environ.get("DV3_BENCH_EAGER_TIMING", "") != "1":
            sys.stderr.write("roofline timing: graph capture failed (%s: %s); eager launches\n" % (type(e).__name__, e))
        torch.cuda.synchronize()
    e0.record(s)
    for _ in range(iters):
        launch()
    e1.record(s)
    torch.cuda.synchronize()
    _time_launches.last_mode = "eager launches"
    return e0.elapsed_time(e1) * 1e3 / iters


_time_launches.last_mode = None


def _traffic(kernel_key):
    """HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, calibrated as MI355X_MICROARCH.md
    prescribes), collected offline with rocprofv3 (scripts/pmc_hbm.sh; a counter pass can not run inside this
    process) and committed under profiles/ -- used only when the file was collected for THIS kernel."""
    for name in ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json"):      # newest collection that holds THIS variant
        try:
            hb = json.load(open(os.path.join(ROOT, "profiles", name)))
            ent = hb[kernel_key]
            return ent["hbm_bytes_per_launch"], "profiles/%s (%s)" % (name, ent["source"])
        except (IOError, OSError, KeyError, ValueError):
            continue
    return None, None


def conv_roofline(dev, iters=100, tile_hint=0, dil=1, mode=None, c8=False):
    """Conv1dGLU forward at the north-star shape, one tap-GEMM launch per iteration.  c8: the bf16-storage form
    (bf16 GEMM mode, channel-blocked bf16 input / residual / output)."""
    from deepvoice3_pytorch_amd import ops, _lib
    mode = "bf16" if c8 else (mode or ops.gemm_precision())
    prev = ops.set_gemm_precision(mode)
    B, C, T, k, d = 64, 256, 1024, 3, dil
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
    g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
    bias = torch.zeros(2 * C, device=dev)
    pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
    y = torch.empty(B, C, T, device=dev)

    x8 = ops.to_c8(x) if c8 else None

    def launch():
        if c8:
            ops.conv_gemm(None, None, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d,
                          padL=(k - 1) // 2 * d, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x8, residual=1,
                          a_split=pk.fwd_s, x_c8=x8, out_c8=True)
        else:
            ops.conv_gemm(x, pk.fwd, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=d,
                          padL=(k - 1) // 2 * d, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x, residual=1, y=y,
                          tile_hint=tile_hint, a_split=pk.fwd_s)
    us = _time_launches(launch, iters)
    variant = _lib.lib().dv3_debug_get(10)
    ops.set_gemm_precision(prev)
    flops = 2.0 * B * T * (2 * C) * (k * C)                     # 51.54 GFLOP
    byts = 4.0 * (B * C * T * 2 + 2 * C * C * k + 2 * C)         # x + y + weights + bias: 135.8 MB
    if c8:
        byts = 2.0 * (B * C * T * 2) + 2.0 * 2 * C * C * k + 4.0 * 2 * C   # bf16 x + y, bf16 weight image, fp32 bias
    tf = flops / (us * 1e-6) / 1e12
    fam = variant // 1000
    peak = {1: PEAK_F32_MFMA_TF, 2: PEAK_F32_MFMA_TF, 4: PEAK_16BIT_MFMA_TF,
            8: PEAK_16BIT_MFMA_TF, 9: PEAK_16BIT_MFMA_TF}.get(fam, PEAK_16BIT_MFMA_TF / 3.0)
    kname = {1: "conv_gemm_f32_stream_kernel", 2: "conv_gemm_f32_kernel", 3: "conv_gemm_bf16x3_kernel<bf16 hi/lo>",
             4: "conv_gemm_bf16x3_kernel<bf16 x1>", 5: "conv_gemm_bf16x3_kernel<fp16 hi/lo>",
             6: "conv_planes_kernel<fp16 hi/lo>", 7: "conv_planes_kernel<bf16>",
             8: "conv_planes_kernel<bf16 x1, c8 storage>",
             9: "conv_c8pp_kernel<bf16 x1, c8 storage, 256 x 256 k32 ping-pong>"}.get(fam, "?")
    key = "conv_fwd:%d" % variant
    if variant % 1000 == 101 and fam in (3, 5):
        kname = "conv_gemm_pp2_kernel<%s>" % ("fp16 hi/lo" if fam == 5 else "bf16 hi/lo")
    out = dict(bound="mfma", kernel="%s variant %d (Conv1dGLU fwd B=64 C=256 T=1024 k=3)" % (kname, variant),
               achieved=round(tf, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(tf / peak, 4),
               traffic=None, us_per_launch=round(us, 2), alg_flops=flops, alg_bytes=byts,
               hbm_gbs=round(byts / (us * 1e-6) / 1e9, 1),
               hbm_frac=round(byts / (us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4), variant=variant,
               timing=_time_launches.last_mode)
    if dil == 1 and tile_hint == 0:
        out["traffic"], src = _traffic(key)
        if src:
            out["traffic_source"] = src
    if fam in (3, 5, 6):
        out["peak_note"] = ("algorithmic fp32 FLOPs against the dense 16-bit MFMA peak (2500 TF) / 3: the split "
                            "kernels issue 3 MFMAs per product block; executed MFMA rate = 3 x achieved")
        out["mfma_executed_tflops"] = round(3 * tf, 1)
        out["x_fp32_matrix_peak"] = round(tf / PEAK_F32_MFMA_TF, 3)
        # informational: what the matrix pipes alone sustain on RANDOM operands on this part (they reach the nominal
        # 2.46 PF only on zeros: scripts/ubench/mfma_power.hip, profiles/r03_mfma_power.txt) -- `frac` above stays
        # priced against the nominal dense peak
        out["frac_of_mfma_rate_on_random_operands"] = round(3 * tf / MFMA_RANDOM_OPERANDS_TF, 3)
        out["mfma_rate_on_random_operands_tflops"] = MFMA_RANDOM_OPERANDS_TF
    return out


def wgrad_roofline(dev, iters=50, mode=None):
    """Weight gradient of the same Conv1dGLU layer (B=64, M=512 gradient rows, Cin=256, T=1024, k=3,
    dropout-masked input as in training): out[j][m][c] = sum_{b,t} g[b][m][t] x[b][c][t+j-1]."""
    from deepvoice3_pytorch_amd import ops, _lib
    mode = mode or ops.gemm_precision()
    prev = ops.set_gemm_precision(mode)
    B, C, T, k = 64, 256, 1024, 3
    torch.manual_seed(0)
    x = torch.randn(B, C, T, device=dev)
    gm = torch.randn(B, 2 * C, T, device=dev)
    bits, rs = ops.dropout_bits(B * C, T, 0.05, dev)
    x3 = mode != "f32"
    tiles = ((2 * C + 127) // 128) * ((C + 127) // 128) * k
    # as ops.ConvLayerFn.backward sizes it: one 8-wave workgroup per (tile, slab) serves the three taps
    S = ops._ksplit_count(B * ((T + 31) // 32), tiles // k, slots=256) if x3 else ops._slab_count(B, tiles)
    rows = bool(ops.slab_rows_default and S > 1)       # the K-split partial sums as the step lays them out (round 4)
    out_t = torch.empty((k, 2 * C, S, C) if rows else (S, k, 2 * C, C), dtype=torch.float32, device=dev)

    # as the training step hands it over since round 6: the pre-gate gradient as pair words (ops.pair_words), which the
    # three-term kernels stage without conversion
    gp = bool(x3 and ops.pair_words)
    if gp:
        gm = ops.pair_words_of(gm)

    def launch():
        ops.wgrad_gemm(gm, x, B=B, M=2 * C, Cin=C, T=T, Tin=T, J=k, dil=1, padL=1, n_slabs=S, xmask=bits,
                       xmask_rs=rs, drop_scale=1.0 / 0.95, split_bf16=x3, k_split=x3, out=out_t, rows_of_slabs=rows,
                       g_pair=gp)
    us = _time_launches(launch, iters, settle=50)
    variant = _lib.lib().dv3_debug_get(11)
    ops.set_gemm_precision(prev)
    flops = 2.0 * B * T * (2 * C) * (k * C)
    byts = 4.0 * (B * 2 * C * T + B * C * T + k * 2 * C * C)         # g + x + dW (the split-K slabs are an artefact)
    tf = flops / (us * 1e-6) / 1e12
    fam = variant // 1000
    peak = {1: PEAK_F32_MFMA_TF, 4: PEAK_16BIT_MFMA_TF}.get(fam, PEAK_16BIT_MFMA_TF / 3.0)
    out = dict(bound="mfma", kernel="wgrad variant %d (Conv1dGLU wgrad B=64 M=512 Cin=256 T=1024 k=3, %d K-slabs%s)" % (
                   variant, S, " as rows [J][M][S][C]" if rows else ""),
               achieved=round(tf, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(tf / peak, 4),
               traffic=None, us_per_launch=round(us, 2), alg_flops=flops, alg_bytes=byts,
               hbm_gbs=round(byts / (us * 1e-6) / 1e9, 1), variant=variant)
    out["traffic"], src = _traffic("wgrad:%d" % variant)
    if src:
        out["traffic_source"] = src
    return out


# -------------------------------------------------------------------------------------------------
# CPU baseline
# -------------------------------------------------------------------------------------------------
def _bench_items(rng, B, Tt, n_frames, hp):
    """ragged items (text ids, mel, linear) of the fixed bench shape, as the reference's datasets yield them"""
    items = []
    for _ in range(B):
        text = np.concatenate([rng.randint(2, hp["n_vocab"], Tt - 1), [1]]).astype(np.int64)
        items.append((text, rng.rand(n_frames, hp["mel_dim"]).astype(np.float32),
                      rng.rand(n_frames, hp["linear_dim"]).astype(np.float32)))
    return items


def cpu_baseline_reference(B, Tt, n_frames, max_seconds=25.0):
    """The UNMODIFIED reference's train.train() (train.py:604-785) on the host cores, imported through
    oracle/refimport.py (stubs for absent non-arithmetic imports; numba.jit = identity, so guided_attention
    runs as plain Python like any install without numba).  Only possible where the reference tree exists."""
    from oracle import refimport
    train, hparams, Writer = refimport.load_train_module()
    import deepvoice3_pytorch.frontend as fe
    hparams.parse_json(open(os.path.join(refimport.REF_ROOT, "presets", "deepvoice3_ljspeech.json")).read())
    train._frontend = fe.en
    torch.manual_seed(0)
    model = train.build_model()
    rng = np.random.RandomState(1234)
    items = [(t.astype(np.int32), m, y) for t, m, y in _bench_items(rng, B, Tt, n_frames, DV3_LJ)]
    batch = train.collate_fn(items)
    opt = torch.optim.Adam(model.get_trainable_parameters(), lr=hparams.initial_learning_rate,
                           betas=(hparams.adam_beta1, hparams.adam_beta2), eps=hparams.adam_eps,
                           weight_decay=hparams.weight_decay, amsgrad=hparams.amsgrad)

    def run(n):
        train.global_step, train.global_epoch = 0, 0
        t0 = time.time()
        train.train(torch.device("cpu"), model, [batch] * n, opt, Writer(), init_lr=hparams.initial_learning_rate,
                    checkpoint_dir="/tmp", checkpoint_interval=10 ** 9, nepochs=1, clip_thresh=hparams.clip_thresh)
        return time.time() - t0
    t1 = run(1)                                   # warm-up step (allocator, thread pool)
    n = int(max(1, min(20, max_seconds // max(t1, 1e-3))))
    dt = run(n) / n
    frames = float(B * n_frames)
    return dict(value=round(frames / dt, 1), unit="mel-frames/s", cores=torch.get_num_threads(), kind="reference",
                sample="%d steps of the reference's own train.train() on the same workload (B=%d, Tt=%d, %d "
                       "frames/item), %.2f s/step" % (n, B, Tt, n_frames, dt), host_cpus=os.cpu_count())


class _PortStep(object):
    """the oracle port of one train step on the host (state for one batch size)"""

    def __init__(self, B, Tt, n_frames):
        from oracle import dv3_oracle as O
        from deepvoice3_pytorch_amd import builder
        self.O = O
        hp = dict(DV3_LJ)
        self.spec = O.build_spec("deepvoice3", **hp)
        torch.manual_seed(0)
        sd = {k: v.detach().clone() for k, v in builder.deepvoice3(**hp).state_dict().items()}
        frozen = ("seq2seq.decoder.embed_query_positions.weight", "seq2seq.decoder.embed_keys_positions.weight")
        self.names = [k for k in sd if k not in frozen]
        for k in self.names:
            sd[k].requires_grad_(True)
        self.sd = sd
        self.m = {k: torch.zeros_like(sd[k]) for k in self.names}
        self.v = {k: torch.zeros_like(sd[k]) for k in self.names}
        rng = np.random.RandomState(1234)
        self.bt = synth_batch(rng, B, Tt, n_frames, hp)
        self.mel = self.bt["mel"][:, 0::4, :].contiguous()
        self.lhp = dict(outputs_per_step=1, downsample_step=4, masked_loss_weight=0.5, binary_divergence_weight=0.1,
                        use_guided_attention=True, guided_attention_sigma=0.2)
        self.frames = float(self.bt["target_lengths"].sum())
        self.it = 0

    @staticmethod
    def _drop(site, t, p, layout):     # F.dropout stand-in with the same cost profile (bernoulli_ + mul)
        return torch.nn.functional.dropout(t, p, True)

    def one(self):
        O, sd, bt, names = self.O, self.sd, self.bt, self.names
        for k in names:
            sd[k].grad = None
        out = O.model_forward(sd, self.spec, bt["text"], self.mel, None, bt["text_positions"], bt["frame_positions"],
                              bt["input_lengths"], drop=self._drop)
        loss, _ = O.train_losses(self.spec, self.lhp, out, self.mel, bt["y"], bt["done"], bt["input_lengths"],
                                 bt["target_lengths"])
        loss.backward()
        self.it += 1
        with torch.no_grad():
            O.clip_and_adam([sd[k] for k in names], [sd[k].grad for k in names], [self.m[k] for k in names],
                            [self.v[k] for k in names], self.it, 5e-4)

    def time(self, max_steps, max_seconds):
        self.one()                      # warm-up at the current thread count (allocator, thread pool)
        t0, n = time.time(), 0
        while n < max_steps and (time.time() - t0) < max_seconds:
            self.one()
            n += 1
        return (time.time() - t0) / n, n


def cpu_baseline_port(B, Tt, n_frames, max_seconds=25.0):
    """The oracle port of the reference train step on the host cores (bounded sample).  The intra-op thread count
    is tuned first: a sweep over {8, 16, 32, 64, 128, all} on a batch-8 slice of the workload (one step each),
    then the full-batch sample runs at the best count."""
    ncpu = os.cpu_count() or 8
    prev_threads = torch.get_num_threads()
    sweep = {}
    try:
        small = _PortStep(min(B, 8), Tt, n_frames)
        for nt in sorted(set(t for t in (8, 16, 32, 64, 128, prev_threads) if t <= ncpu)):
            torch.set_num_threads(nt)
            dt_s, _ = small.time(1, 6.0)
            sweep[nt] = round(small.frames / dt_s, 1)
        best = max(sweep, key=sweep.get)
        del small
        torch.set_num_threads(best)
        full = _PortStep(B, Tt, n_frames)
        dt, n = full.time(20, max_seconds)
        frames = full.frames
    finally:
        torch.set_num_threads(prev_threads)
    out = dict(value=round(frames / dt, 1), unit="mel-frames/s", cores=best, kind="port",
               sample="%d train steps of the same workload (B=%d, Tt=%d, %d frames/item) through "
                      "oracle/dv3_oracle.py on the host at the best of the swept thread counts, %.2f s/step"
                      % (n, B, Tt, n_frames, dt),
               host_cpus=ncpu, thread_sweep_frames_per_s_at_batch8=sweep)
    try:      # port / reference time ratio recorded once in the build container (scripts/cpu_baseline_calibration.py)
        # round 5: re-measured at the headline batch (B = 64 and 16, all cores and one thread, the share of the reference's
        # pure-Python guided_attention timed separately): the port takes 0.80 of the reference's step time at B = 64 on
        # 8 cores (round 2's 0.185 was a B = 8 measurement, where guided_attention and Python overhead dominate)
        cal = json.load(open(os.path.join(ROOT, "profiles", "r05_cpu_port_vs_reference.json")))
        out["port_over_reference_time"] = cal["port_over_reference_time"]
        out["reference_guided_attention_share"] = [r for r in cal["runs"] if r["batch"] == 64][0]["guided_attention_share_of_reference_step"]
        out["calibration"] = "profiles/r05_cpu_port_vs_reference.json (%s, %d cores)" % (cal.get("cpu_model", "?"), cal.get("host_cpus", 0))
    except (IOError, OSError, KeyError, ValueError):
        pass
    return out


def cpu_baseline(B, Tt, n_frames, max_seconds=25.0):
    from oracle import refimport
    if refimport.available():
        try:
            return cpu_baseline_reference(B, Tt, n_frames, max_seconds)
        except Exception as e:      # the reference tree is there but does not import here: say so, use the port
            sys.stderr.write("reference cpu baseline failed (%s: %s); using the oracle port\n" % (type(e).__name__, e))
    return cpu_baseline_port(B, Tt, n_frames, max_seconds)


# -------------------------------------------------------------------------------------------------
# synthesis (BASELINE.json configs[4])
# -------------------------------------------------------------------------------------------------
def synth_run(dev, batch=64, reps=3, warm=1, gl_iters=60, step_graph=True):
    """synthesis.py's path (synthesis.py:42-73) for `batch` concurrent utterances: greedy autoregressive decode
    (Decoder.incremental_forward) + Converter + Griffin-Lim vocoder on the device.  SURVEY.md 8(d) cfg5: equal
    text length 100, min = max decoder steps = 200 -> 201 steps = 804 frames = 9.33 s of audio each;
    RTF = wall / audio seconds."""
    from deepvoice3_pytorch_amd import builder, audio, ops
    hp = dict(DV3_LJ)
    torch.manual_seed(0)
    model = builder.deepvoice3(**hp).to(dev).eval()
    model.make_generation_fast_()
    dec = model.seq2seq.decoder
    dec.min_decoder_steps = dec.max_decoder_steps = 200
    dec.use_step_graph = step_graph       # replay one hipGraph per decoder step
    B, Tt = batch, 100
    rng = np.random.RandomState(0)
    text = torch.from_numpy(rng.randint(2, hp["n_vocab"], (B, Tt))).to(dev)
    tpos = torch.arange(1, Tt + 1).repeat(B, 1).to(dev)
    acfg = audio.AudioConfig(griffin_lim_iters=gl_iters)

    def run():
        t = [time.perf_counter()]
        with torch.no_grad():
            mel, lin, align, done = model(text, text_positions=tpos)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        wav = audio.inv_spectrogram_batch(lin, acfg)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        return lin, wav, t
    for _ in range(warm):
        run()
    tm, tv = [], []
    for _ in range(reps):
        lin, wav, t = run()
        tm.append(t[1] - t[0]); tv.append(t[2] - t[1])
    assert torch.isfinite(wav).all() and lin.shape[1] == 804
    audio_s = B * wav.shape[1] / acfg.sample_rate
    wall = float(np.mean(tm) + np.mean(tv))
    return dict(value=round(wall / audio_s, 6), unit="wall seconds per audio second", higher_is_better=False,
                ms_per_step=round(wall * 1e3, 2), reps=reps, dtype=ops.gemm_precision(),
                config=dict(workload="builder=deepvoice3 preset=deepvoice3_ljspeech synthesis, Tt=100, 201 decoder steps "
                                     "= 804 frames per utterance", utterances=B, griffin_lim_iters=gl_iters,
                            decode_loop=("persistent program (one launch)" if os.environ.get("DV3_DECODE_PERSISTENT") == "1"
                                         else "python launches" + (", per-step hipGraph" if dec.use_step_graph else "")
                                         if os.environ.get("DV3_DECODE_LAUNCHED") == "0"
                                         else "library-launched chunks of steps (dv3_decode_program_launch)"),
                            audio_seconds=round(audio_s, 1),
                            model_ms=round(float(np.mean(tm)) * 1e3, 1), vocoder_ms=round(float(np.mean(tv)) * 1e3, 1),
                            rtf_model_only=round(float(np.mean(tm)) / audio_s, 6),
                            ms_per_decoder_step=round(float(np.mean(tm)) * 1e3 / 201, 3)))


# -------------------------------------------------------------------------------------------------
# the train step
# -------------------------------------------------------------------------------------------------
class _PowerSampler(object):
    """Shader clock and socket power of one GPU sampled from sysfs by a thread while the timed loop runs (the step
    runs at the chip's power limit: profiles/r02c; the guide's "DVFS give-back").  Best effort: None when the
    node does not expose the files."""

    def __init__(self, index, period=0.02):
        import glob
        import threading
        self.sclk, self.power = [], []
        base = None
        for card in sorted(glob.glob("/sys/class/drm/card*/device")):
            if os.path.exists(os.path.join(card, "pp_dpm_sclk")):
                if index == 0:
                    base = card
                    break
                index -= 1
        self.f_sclk = os.path.join(base, "pp_dpm_sclk") if base else None
        hw = sorted(glob.glob(os.path.join(base, "hwmon", "hwmon*"))) if base else []
        self.f_pow = None
        for h in hw:
            for name in ("power1_average", "power1_input"):
                if os.path.exists(os.path.join(h, name)):
                    self.f_pow = os.path.join(h, name)
                    break
        self.period, self._stop = period, False
        self.th = None
        if self.f_sclk or self.f_pow:
            self.th = threading.Thread(target=self._run, daemon=True)
            self.th.start()

    def _run(self):
        while not self._stop:
            try:
                if self.f_sclk:
                    for line in open(self.f_sclk).read().splitlines():
                        if line.rstrip().endswith("*"):
                            self.sclk.append(float(line.split(":")[1].strip().split("M")[0]))
                if self.f_pow:
                    self.power.append(float(open(self.f_pow).read()) * 1e-6)
            except (IOError, OSError, ValueError, IndexError):
                pass
            time.sleep(self.period)

    def stop(self):
        if self.th is None:
            return None
        self._stop = True
        self.th.join(1.0)
        out = dict(samples=max(len(self.sclk), len(self.power)), source="sysfs pp_dpm_sclk / hwmon power1_average")
        if self.sclk:
            out["avg_sclk_mhz"] = round(sum(self.sclk) / len(self.sclk), 1)
            out["min_sclk_mhz"] = min(self.sclk)
        if self.power:
            out["avg_power_w"] = round(sum(self.power) / len(self.power), 1)
            out["max_power_w"] = round(max(self.power), 1)
        return out


class TrainRun(object):
    """model + trainer + resident batch of one (preset, gemm mode); .measure() = the contract's timed loop"""

    def __init__(self, dev, pg, rank, world, preset, gemm, batch, text_len, frames, graph, ragged=False, standin=None,
                 trainer_kw=None):
        """standin: a dist.RingStandin in the process group's place (one GPU, no group): the data-parallel step's bucket
        schedule with a measurement stand-in for the ring kernels (configs.ddp_standin)"""
        from deepvoice3_pytorch_amd import builder, train_step, ops
        self.ops, self.train_step = ops, train_step
        self.prev_mode = ops.set_gemm_precision(gemm)
        self.dev, self.pg, self.rank, self.world = dev, pg, rank, world
        self.preset, self.gemm = preset, gemm
        bname, hp0, ga_sigma = PRESETS[preset]
        self.bname, self.hp = bname, dict(hp0)
        torch.manual_seed(0)            # identical initial weights on every rank
        self.model = getattr(builder, bname)(**self.hp).to(dev)
        cfg = train_step.TrainConfig(max_positions=self.hp["max_positions"], guided_attention_sigma=ga_sigma)
        self.trainer = train_step.Trainer(self.model, cfg, process_group=standin if standin is not None else pg,
                                          **(trainer_kw or {}))
        rng = np.random.RandomState(1234 + rank)
        self.bt = synth_batch(rng, batch, text_len, frames, self.hp, fixed=not ragged)
        self.spk = torch.from_numpy(rng.randint(0, self.hp["n_speakers"], batch)) if self.hp["n_speakers"] > 1 else None
        bt = self.bt
        self.batch = train_step.Batch.from_collate(bt["text"], bt["input_lengths"], bt["mel"], bt["y"],
                                                   bt["text_positions"], bt["frame_positions"], bt["done"],
                                                   bt["target_lengths"], self.spk, downsample_step=4, device=dev)
        self.trainer.check_lengths(self.batch)
        # Launch mode.  The step is ~300-390 kernel launches: 5-16 ms of Python + ctypes per step for eager launches
        # (box dependent) against 1.5-4 ms for a replay.  A single whole-step hipGraph serialises the two backward
        # branches (round 3: 4-7 % slower than eager whenever the GPU is the bound), so since round 4 the replay is
        # a chain of SEGMENT graphs -- step stream | weight-gradient branch on the real second stream | optimiser
        # (train_step.GraphedTrainer(split_streams), include/dv3hip.h: dv3_graph_fork) -- which runs at the eager
        # step's GPU time (profiles/r04_three_graph_probe.txt).  The default still PROBES eager against the replay for
        # a few steps and keeps the faster; --graph / --no-graph force one.  All ranks take the same decision (MAX).
        self.runner = None
        self.use_graph = False
        self.graph_error = None
        self.launch_probe = None
        t_eager = None
        if graph == "auto":
            t_eager = self._probe(4)
        self.use_graph = bool(graph)
        if self.use_graph:
            ok = 1
            if t_eager is not None and os.environ.get("DV3_BENCH_EMPTY_CACHE", "1") != "0":
                # the eager probe leaves its blocks cached in the default pool; the capture allocates from a private one:
                # hand the memory back first (measured: the deepvoice3_vctk replay captured after an eager probe ran 6 %
                # slower than the same replay captured in a fresh process state)
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
            try:
                self.runner = train_step.GraphedTrainer(self.trainer, self.batch, warmup=2)
            except Exception as e:      # capture not possible: say so, go eager
                ok = 0
                self.graph_error = "%s: %s" % (type(e).__name__, e)
                if rank == 0:
                    import traceback
                    traceback.print_exc()
                    print("hipGraph capture failed (%s); running eager" % type(e).__name__, file=sys.stderr)
                torch.cuda.synchronize()
            if pg is not None:          # replay only if EVERY rank captured (a mixed job would dead-lock its collectives)
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                ok = int(flag.item())
            if not ok:
                if self.runner is not None:
                    self.runner.close()
                self.runner, self.use_graph = None, False
        if graph == "auto" and self.use_graph:
            t_graph = self._probe(4)
            self.launch_probe = dict(eager_ms_per_step=round(t_eager, 3), hipgraph_ms_per_step=round(t_graph, 3), steps=4)
            # the replay has the eager step's GPU time since round 4 (segment graphs) and a tenth of its host
            # cost, so a tie goes to the replay: it stays GPU-bound on a slow or busy host; eager only when clearly faster
            if t_graph > 1.01 * t_eager:
                self.runner.close()
                self.runner, self.use_graph = None, False
                torch.cuda.synchronize()
                torch.cuda.empty_cache()

        # The flag-ordered backward (train_step.GraphedTrainer.flag_sync, ABI 43) is opt-in in the library; where its rule
        # applies (fp32 activations below batch 48, no process group, a second stream seen beside the step stream) it is
        # probed here against the segments the same way, and kept only when it is the faster form IN THIS PROCESS.
        if (self.use_graph and self.runner is not None and getattr(self.runner, "split", False) and pg is None
                and os.environ.get("DV3_FLAG_SYNC") is None and int(self.batch.mel.size(0)) < 48
                and not ops.storage_c8() and getattr(self.trainer, "side_stream_beside", False)):
            t_seg = self._probe(4)
            seg_runner = self.runner
            seg_runner.close()
            self.runner = None
            os.environ["DV3_FLAG_SYNC"] = "1"
            try:
                self.runner = train_step.GraphedTrainer(self.trainer, self.batch, warmup=2)
                t_flag = self._probe(4)
                bad = self.runner.flag_timeouts()
            except Exception as e:
                t_flag, bad = float("inf"), -1
                if rank == 0:
                    print("flag-ordered capture failed (%s: %s); keeping the segments" % (type(e).__name__, e), file=sys.stderr)
            finally:
                del os.environ["DV3_FLAG_SYNC"]
            if self.launch_probe is None:      # (--graph: no eager probe was run)
                self.launch_probe = dict(eager_ms_per_step=None, hipgraph_ms_per_step=round(t_seg, 3), steps=4)
            self.launch_probe.update(segments_ms_per_step=round(t_seg, 3),
                                     flag_ordered_ms_per_step=(round(t_flag, 3) if math.isfinite(t_flag) else None))
            if not (t_flag < 0.995 * t_seg and bad == 0):
                if self.runner is not None:
                    self.runner.close()
                self.runner = train_step.GraphedTrainer(self.trainer, self.batch, warmup=2)

    def graph_form(self):
        if not self.use_graph or self.runner is None:
            return None
        if getattr(self.runner, "flag_sync", False):
            return ("forward graph | backward graph on the step stream beside ONE weight-gradient graph on the second stream, "
                    "their %d fork points ordered by a device flag (dv3_flag_signal / dv3_flag_wait, ABI 43) | optimiser graph"
                    % self.runner.n_forks)
        if getattr(self.runner, "split", False):
            return ("%d segment hipGraphs on the step stream, each followed by its weight-gradient segment on the second "
                    "stream (and, under a process group, by the host-issued all-reduces of the buckets it completes) | "
                    "optimiser graph" % len(self.runner.segs))
        return "one hipGraph"

    def _probe(self, n):
        """ms per step of the current launch mode over n steps (after 2 untimed ones), MAX over ranks"""
        for _ in range(2):
            self.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            self.step()
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / n * 1e3], dtype=torch.float64, device=self.dev)
        if self.pg is not None:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    def step(self, batch=None):
        if self.use_graph:
            return self.runner.step(batch)      # a fresh batch is copied into the captured one first
        return self.trainer.step(batch if batch is not None else self.batch)

    def measure(self, steps, warmup, feed=None, settle_s=0.0):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronize; max over ranks.
        feed: an iterator of device batches (data.Prefetcher) used instead of the resident batch."""
        pg, dev = self.pg, self.dev
        nxt = (lambda: next(feed)) if feed is not None else (lambda: None)
        comm = self.trainer.comm
        if settle_s > 0:
            # the clock governor needs ~1 s of this load before the step time is stationary; a short --warmup
            # would otherwise time the transient.  Same count on every rank (collectives inside the step).
            t_s = time.perf_counter()
            for _ in range(2):
                scal = self.step(nxt())
            torch.cuda.synchronize()
            per = max((time.perf_counter() - t_s) / 2, 1e-4)
            n_settle = torch.tensor([int(min(200, max(0, settle_s / per - 2)))], dtype=torch.int32, device=dev)
            if pg is not None:
                torch.distributed.all_reduce(n_settle, op=torch.distributed.ReduceOp.MAX)
            for _ in range(int(n_settle.item())):
                scal = self.step(nxt())
        for _ in range(warmup):
            scal = self.step(nxt())
        if comm is not None:
            # measurable whenever the collectives are issued from the host: eager, and the segmented replay
            comm.exposed_events = [] if (not self.use_graph or getattr(self.runner, "split", False)) else None
        if pg is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        sampler = _PowerSampler(dev.index or 0) if self.rank == 0 else None
        t0 = time.perf_counter()
        for _ in range(steps):
            scal = self.step(nxt())
        t_host = time.perf_counter() - t0       # all K steps enqueued; the GPU is still working if it is the bound
        torch.cuda.synchronize()
        if pg is not None:
            torch.distributed.barrier()
        dt = time.perf_counter() - t0
        power = sampler.stop() if sampler is not None else None
        # Host time to issue ONE step into an idle queue.  (t_host above also contains back-pressure: once the host
        # is a few steps ahead the runtime makes it wait -- a replayed hipGraph cannot be launched again while it is
        # in flight, and eager launches block when the queue's kernarg ring is full -- so over a long loop it
        # converges to the GPU's step time whatever the host really costs.)
        t_issue = 0.0
        for _ in range(3):
            torch.cuda.synchronize()
            if pg is not None:
                torch.distributed.barrier()
            h0 = time.perf_counter()
            scal = self.step(nxt())
            t_issue += time.perf_counter() - h0
        torch.cuda.synchronize()
        exposed = comm.exposed_ms() if (comm is not None and comm.exposed_events is not None) else None
        if comm is not None:
            comm.exposed_events = None
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        frames = torch.tensor([float(self.batch.n_frames)], dtype=torch.float64, device=dev)
        if pg is not None:
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            torch.distributed.all_reduce(frames, op=torch.distributed.ReduceOp.SUM)
        dt = float(tmax.item())
        loss = float(scal["loss"])
        if not math.isfinite(loss):
            raise RuntimeError("non-finite training loss (%r): the measurement is void" % loss)
        ms = dt / steps * 1e3
        value = float(frames.item()) / (dt / steps)
        tf = MFLOP_PER_FRAME[self.preset] * 1e6 * value / 1e12
        return dict(value=round(value, 1), ms_per_step=round(ms, 3), steps=steps, warmup=warmup,
                    host_enqueue_ms_per_step=round(t_issue / 3 * 1e3, 3),
                    host_loop_ms_per_step=round(t_host / steps * 1e3, 3), allreduce_exposed_ms=exposed, power=power,
                    final_loss=round(loss, 5), frames_per_step=float(frames.item()),
                    step_flop_frac=dict(alg_mflop_per_frame=MFLOP_PER_FRAME[self.preset], achieved_tflops=round(tf, 1),
                                        peak=round(mfma_peak_tf(self.gemm), 1),
                                        frac=round(tf / mfma_peak_tf(self.gemm), 4)))

    def close(self):
        if self.runner is not None:
            self.runner.close()
        self.trainer.close()
        self.ops.dropout_state.dev_offset = None
        self.ops.set_gemm_precision(self.prev_mode)
        self.runner = self.trainer = self.model = self.batch = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


def launch_mode(args, world=1):
    """--graph / --no-graph force the launch mode; the default PROBES both (TrainRun) and keeps the faster step, on every
    world size: the segmented replay (train_step.GraphedTrainer) captures nothing of the process group -- its
    all-reduces are ordinary c10d calls issued from the host between segment launches, in an order fixed by the model
    -- so a multi-rank replay runs the same collective code path as the eager step.  All ranks take the same decision
    (a MIN over "did my capture succeed", a MAX over the probe times).  DV3_BENCH_DDP_GRAPH=0 keeps more than one rank
    eager without probing."""
    if args.no_graph:
        return False
    if args.graph:
        return True
    if world > 1 and os.environ.get("DV3_BENCH_DDP_GRAPH", "auto") in ("0", "eager"):
        return False
    return "auto"


def side_config(dev, pg, rank, world, preset, gemm, args, steps, warmup, batch=None, ragged=False):
    """one of the secondary configurations of the line.  A failure here (a capture that does not fit, a non-finite loss)
    is RECORDED in the entry -- it must not cost the run its headline -- except under a process group, where every rank
    has to take the same path."""
    if pg is not None:
        return _side_config(dev, pg, rank, world, preset, gemm, args, steps, warmup, batch, ragged)
    try:
        return _side_config(dev, pg, rank, world, preset, gemm, args, steps, warmup, batch, ragged)
    except Exception as e:
        import traceback
        traceback.print_exc()
        return dict(error="%s: %s" % (type(e).__name__, e), preset=preset, dtype=gemm)


def _side_config(dev, pg, rank, world, preset, gemm, args, steps, warmup, batch=None, ragged=False):
    batch = batch or args.batch
    run = TrainRun(dev, pg, rank, world, preset, gemm, batch, args.text_len, args.frames,
                   graph=launch_mode(args, world), ragged=ragged)
    try:
        m = run.measure(steps, warmup)
        used_graph, probe, gform = bool(run.use_graph), run.launch_probe, run.graph_form()
        shape = dict(text_len=int(run.bt["text"].shape[1]), padded_frames=int(run.bt["mel"].shape[1]),
                     frames_per_step=m["frames_per_step"])
    finally:
        run.close()
    return dict(metric="mel-frames/sec/node (train step, %s)" % preset, value=m["value"], unit="mel-frames/s",
                n_gpus=world, steps=steps, warmup=warmup, ms_per_step=m["ms_per_step"], dtype=gemm,
                dtype_note=dtype_note(gemm), step_flop_frac=m["step_flop_frac"],
                config=dict(workload="builder=%s preset=%s train step" % (PRESETS[preset][0], preset),
                            per_gpu_batch=batch, global_batch=batch * world, final_loss=m["final_loss"],
                            lengths=("ragged LJSpeech-shaped (SURVEY 8d cfg2: frames ~ clip(N(566,180),120,870), text ~ "
                                     "clip(N(100,30),20,187)), padded to the batch maximum as train.collate_fn does"
                                     if ragged else "fixed"), shape=shape,
                            hipgraph=used_graph, graph_form=gform, launch_probe=probe,
                            host_enqueue_ms_per_step=m["host_enqueue_ms_per_step"],
                            host_loop_ms_per_step=m["host_loop_ms_per_step"],
                            launch_bound=bool(m["host_enqueue_ms_per_step"] > 0.97 * m["ms_per_step"]),
                            host_below_half_step=bool(m["host_enqueue_ms_per_step"] < 0.5 * m["ms_per_step"]),
                            allreduce_exposed_ms=m["allreduce_exposed_ms"]))


def ragged_epoch_config(dev, preset, gemm, args, n_items=13100, n_batches=10):
    """What an epoch over LJSpeech-shaped data costs: `n_items` item lengths from SURVEY 8d
    cfg2's distribution, cut into mini-batches by data.LengthBucketedSampler -- the reference's
    PartialyRandomizedSimilarTimeLengthSampler (train.py:195-239): sorted by length, shuffled inside groups of 32
    batches, so a batch pads to the maximum of SIMILAR lengths -- and `n_batches` of them, spread over the epoch, run as
    optimisation steps.  The padded shape changes every step: no replay (a hipGraph is captured for one shape), eager
    launches, so the host's issue time is part of the number.  value = un-padded frames of those batches / wall time."""
    from deepvoice3_pytorch_amd import data, train_step
    rng = np.random.RandomState(4321)
    frames = np.clip(rng.normal(566, 180, n_items), 120, 870).astype(np.int64)
    text = np.clip(frames * (100.0 / 566.0) + rng.normal(0, 8, n_items), 20, 187).astype(np.int64)
    sampler = data.LengthBucketedSampler(frames, batch_size=args.batch, seed=0)
    batches = [b for b in sampler.epoch_batches() if len(b) == args.batch]
    pick = [batches[i] for i in np.linspace(0, len(batches) - 1, n_batches).astype(int)]
    run = TrainRun(dev, None, 0, 1, preset, gemm, args.batch, args.text_len, args.frames, graph=False)
    try:
        dev_batches, real, padded = [], 0, 0
        for idx in pick:
            bt = synth_batch(rng, len(idx), 0, 0, run.hp, lengths=(text[idx], frames[idx]))
            dev_batches.append(train_step.Batch.from_collate(bt["text"], bt["input_lengths"], bt["mel"], bt["y"],
                                                             bt["text_positions"], bt["frame_positions"], bt["done"],
                                                             bt["target_lengths"], None, downsample_step=4, device=dev))
            real += int(frames[idx].sum())
            padded += int(bt["mel"].shape[0] * bt["mel"].shape[1])
        for b in dev_batches:               # every shape once untimed (allocator, workspaces)
            run.trainer.step(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in dev_batches:
            scal = run.trainer.step(b)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        loss = float(scal["loss"])
        if not math.isfinite(loss):
            raise RuntimeError("non-finite training loss in the ragged epoch")
    finally:
        run.close()
    # the same items in arrival order (no length bucketing): what the padding would be
    naive = [np.arange(i, i + args.batch) for i in np.linspace(0, n_items - args.batch, n_batches).astype(int)]
    pad_naive = sum(int(args.batch * (frames[i].max() + 8)) for i in naive) / float(sum(int(frames[i].sum()) for i in naive))
    return dict(value=round(real / dt, 1), unit="mel-frames/s", ms_per_step=round(dt / len(dev_batches) * 1e3, 3),
                host_enqueue_ms_per_step=round(t_issue / len(dev_batches) * 1e3, 3), steps=len(dev_batches),
                per_gpu_batch=args.batch, hipgraph=False,
                real_frames=real, padded_frames=padded, padded_over_real=round(padded / float(real), 4),
                padded_over_real_without_length_bucketing=round(pad_naive, 4), final_loss=round(loss, 5),
                lengths="%d items, frames ~ clip(N(566,180),120,870); mini-batches by data.LengthBucketedSampler (train.py:195-239), "
                        "%d of the epoch's %d batches, padded to the batch maximum as train.collate_fn does" % (n_items, len(dev_batches), len(batches)))


def ragged_lattice_config(dev, preset, gemm, args, batch, n_items=13100, n_batches=64, lattice=(16, 8)):
    """The ragged epoch WITHOUT the host in the loop: the same item lengths and the same sampler as
    ragged_epoch_config, but every mini-batch is padded to the next point of a lattice of shapes (text positions to a
    multiple of lattice[0], decoder steps to a multiple of lattice[1]) and carries its own maxima as device scalars
    (ops.ValidLengths) -- the step computes what the reference computes on the batch padded to its own maxima
    (tests/test_gpu_valid_lengths.py) -- so its shape is one of a few dozen and train_step.LatticeReplay replays a step
    captured for that shape.  `n_batches` batches spread over the epoch run twice: the first pass meets every shape
    (capture cost reported), the second is timed.  value = un-padded frames of the batches / wall time of the pass."""
    from deepvoice3_pytorch_amd import data, train_step
    rng = np.random.RandomState(4321)
    frames = np.clip(rng.normal(566, 180, n_items), 120, 870).astype(np.int64)
    text = np.clip(frames * (100.0 / 566.0) + rng.normal(0, 8, n_items), 20, 187).astype(np.int64)
    sampler = data.LengthBucketedSampler(frames, batch_size=batch, seed=0)
    batches = [b for b in sampler.epoch_batches() if len(b) == batch]
    pick = [batches[i] for i in np.linspace(0, len(batches) - 1, min(n_batches, len(batches))).astype(int)]
    run = TrainRun(dev, None, 0, 1, preset, gemm, batch, args.text_len, args.frames, graph=False)
    rep = None
    try:
        hp = run.hp
        r, ds = hp["r"], hp["downsample_step"]
        # one pool of random features, sliced per item (the values do not matter, generating 100 batches of them does)
        pool_n = int(batch * 880)
        mel_pool = torch.from_numpy(rng.rand(pool_n, hp["mel_dim"]).astype(np.float32))
        lin_pool = torch.from_numpy(rng.rand(pool_n, hp["linear_dim"]).astype(np.float32))
        dev_batches, real, padded, padded_own = [], 0, 0, 0
        for idx in pick:
            tl, fl = text[idx], frames[idx]
            ids = rng.randint(2, hp["n_vocab"], int(tl.sum())).astype(np.int64)
            ids[np.cumsum(tl) - 1] = 1
            n = int(fl.sum())
            packed = data.PackedBatch(torch.from_numpy(ids), mel_pool[:n], lin_pool[:n], tl.astype(np.int64),
                                      fl.astype(np.int64), None)
            b = data.device_collate(packed, dev, r, ds, lattice=lattice)
            dev_batches.append(b)
            real += int(fl.sum())
            padded += int(b.y.shape[0] * b.y.shape[1])
            padded_own += int(batch * data.padded_frames(fl, r, ds)[0])
        torch.cuda.synchronize()
        rep = train_step.LatticeReplay(run.trainer)
        t0 = time.perf_counter()
        for b in dev_batches:               # first pass: every shape is met (two dry passes + a capture each), then replayed
            rep.step(b)
        torch.cuda.synchronize()
        first_pass_s = time.perf_counter() - t0
        captures, capture_s = rep.stats["captures"], rep.stats["capture_s"]
        t0 = time.perf_counter()
        for b in dev_batches:
            scal = rep.step(b)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        loss = float(scal["loss"])
        if not math.isfinite(loss):
            raise RuntimeError("non-finite training loss in the lattice replay")
        if rep.stats["captures"] != captures:
            raise RuntimeError("the timed pass captured again")
        # host time to issue ONE step into an idle queue (TrainRun.measure: over a long loop the host's time converges to
        # the GPU's through back-pressure, whatever it really costs)
        t_one = 0.0
        probe = dev_batches[:: max(1, len(dev_batches) // 6)][:6]
        for b in probe:
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            rep.step(b)
            t_one += time.perf_counter() - h0
        torch.cuda.synchronize()
    finally:
        if rep is not None:
            rep.close()
        run.close()
    return dict(value=round(real / dt, 1), unit="mel-frames/s", ms_per_step=round(dt / len(dev_batches) * 1e3, 3),
                host_enqueue_ms_per_step=round(t_one / len(probe) * 1e3, 3),
                host_loop_ms_per_step=round(t_issue / len(dev_batches) * 1e3, 3), steps=len(dev_batches),
                per_gpu_batch=batch, hipgraph=True, lattice=dict(text_step=lattice[0], decoder_step=lattice[1]),
                shapes_captured=captures, capture_s_total=round(capture_s, 2),
                capture_s_per_shape=round(capture_s / max(captures, 1), 3),
                capture_amortised_after_steps=int(capture_s / max(dt / len(dev_batches), 1e-9)),
                first_pass_s=round(first_pass_s, 2),
                real_frames=real, padded_frames=padded, padded_over_real=round(padded / float(real), 4),
                padded_over_real_at_batch_maxima=round(padded_own / float(real), 4), final_loss=round(loss, 5),
                lengths="%d items, frames ~ clip(N(566,180),120,870); mini-batches by data.LengthBucketedSampler (train.py:195-239), "
                        "%d of the epoch's %d batches, each padded to the lattice and carrying its own maxima (ops.ValidLengths)"
                        % (n_items, len(dev_batches), len(batches)))


def ddp_world1_config(dev, preset, gemm, args, no_group_ms, steps=12, warmup=4):
    """The data-parallel step with its gradient exchange ARMED, on the one GPU a bench box has: a world-size-1 "nccl"
    (RCCL) group, dist.BucketedAllReduce's notifications, bucketed all-reduces on the collective stream, clip with
    the 1/world prescale.  Measured in both launch modes: eager (two real backward streams + the collective stream)
    and the segmented replay, whose all-reduces are issued from the host between segment launches.  What it shows: the host cost of a step
    with the group armed, and the step time against the same step without a group."""
    import torch.distributed as tdist
    pg = tdist.group.WORLD
    out = dict(no_group_ms_per_step=no_group_ms)
    from deepvoice3_pytorch_amd import ops as _ops
    for mode in ("eager", "hipgraph"):
        n_log = len(_ops.stream_probe_log)
        try:
            run = TrainRun(dev, pg, 0, 1, preset, gemm, args.batch, args.text_len, args.frames, graph=(mode == "hipgraph"))
        except Exception as e:
            out[mode] = dict(error="%s: %s" % (type(e).__name__, e))
            continue
        try:
            if mode == "hipgraph" and not run.use_graph:
                out[mode] = dict(error=run.graph_error or "capture failed")
                continue
            m = run.measure(steps, warmup, settle_s=args.settle)     # same clock-settle time as the headline's measurement
            comm = run.trainer.comm
            out.setdefault("gradient_buckets", len(comm.buckets))
            out.setdefault("bucket_mb", [round((hi - lo) * 4 / 2 ** 20, 2) for lo, hi, _ in comm.buckets])
            out["autograd_hooks_after_first_step"] = len(comm._handles)
            out["parameters"] = len(run.trainer.arena.params)
            out[mode] = dict(ms_per_step=m["ms_per_step"], value=m["value"],
                             host_enqueue_ms_per_step=m["host_enqueue_ms_per_step"],
                             allreduce_exposed_ms=m["allreduce_exposed_ms"],
                             vs_no_group=round(m["ms_per_step"] / no_group_ms, 4) if no_group_ms else None,
                             # which streams the step ran on and what the hardware-queue probe of each found: the ratio of
                             # a pair of spin kernels (candidate beside each avoided stream) to one -- 1.0 runs beside, 2.0
                             # shares a queue; the last candidate listed is the one taken
                             stream_queues=dict(
                                 collectives_issued_from="weight-gradient stream (async)" if comm.async_issue else "own collective stream",
                                 probes=[dict(role=r["role"], found=r["found"], probed=r["probed"], ratios=r["candidates"][-3:])
                                         for r in _ops.stream_probe_log[n_log:]],
                                 GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES", "default (4)")))
        except Exception as e:
            out[mode] = dict(error="%s: %s" % (type(e).__name__, e))
        finally:
            run.close()
    ok = [k for k in ("eager", "hipgraph") if "ms_per_step" in out.get(k, {})]
    if ok:
        out["faster"] = min(ok, key=lambda k: out[k]["ms_per_step"])
    return out


def ddp_standin_config(dev, preset, gemm, batch, args, no_group_ms, world=8, channels=16, threads=256, busbw_gbps=150.0,
                       steps=12, warmup=4, bucket_mb=None):
    """What ONE GPU can measure about the collective of an 8-GPU step: the data-parallel
    step's bucket schedule with dist.RingStandin in the communicator's place -- `channels` persistent workgroups that
    stream 2 (n - 1) / n of every bucket through HBM at the pace an n-rank xGMI ring would set (busbw), issued where the
    bucket all-reduces are issued, awaited where they are awaited.  Reports the stretch of the step beside such a third
    tenant of the CUs, the time the step stream waits for the last buckets, and the 8-GPU weak-scaling efficiency the two
    predict: no_group_ms / standin_ms (every rank does the same work; the collective's wire time is in the stand-in's
    pace).  Segmented replay, like the headline."""
    from deepvoice3_pytorch_amd import dist as _dist
    sd = _dist.RingStandin(world=world, channels=channels, threads=threads, busbw_gbps=busbw_gbps, device=dev)
    kw = dict(bucket_mb=float(bucket_mb)) if bucket_mb is not None else None
    run = TrainRun(dev, None, 0, 1, preset, gemm, batch, args.text_len, args.frames, graph=True, standin=sd, trainer_kw=kw)
    try:
        if not run.use_graph:
            return dict(error=run.graph_error or "capture failed")
        m = run.measure(steps, warmup, settle_s=args.settle)
        comm = run.trainer.comm
        sizes = [(hi - lo) * 4 for lo, hi, _ in comm.buckets]
        wire = [sd.ideal_ms(b) for b in sizes]
        seg = getattr(run.runner, "seg_buckets", None)
        out = dict(ms_per_step=m["ms_per_step"], no_group_ms_per_step=no_group_ms,
                   step_inflation=round(m["ms_per_step"] / no_group_ms, 4) if no_group_ms else None,
                   allreduce_exposed_ms=m["allreduce_exposed_ms"],
                   predicted_weak_scaling_efficiency=round(no_group_ms / m["ms_per_step"], 4) if no_group_ms else None,
                   per_gpu_batch=batch, ranks_modelled=world, channels=channels, threads=threads, busbw_gbps=busbw_gbps,
                   bucket_mb=[round(b / 2 ** 20, 2) for b in sizes],
                   wire_ms_per_bucket=[round(w, 3) for w in wire], wire_ms_per_step=round(sum(wire), 3),
                   buckets_issued_after_segment=[[int(b) for b in bs] for bs in seg] if seg is not None else None,
                   host_enqueue_ms_per_step=m["host_enqueue_ms_per_step"])
        return out
    finally:
        run.close()


def _world1_group():
    """a world-size-1 "nccl" group in this process (no launcher): RCCL loads and runs its collectives on one device"""
    import torch.distributed as tdist
    if tdist.is_initialized():
        return False
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    tdist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
    return True


# -------------------------------------------------------------------------------------------------
# multi-rank launch
# -------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher around it: re-execute this script under
    torch.distributed.run with N ranks on this node (one per GPU, RCCL over xGMI; 127.0.0.1 rendezvous).
    Rank 0 prints the JSON line; the launcher's exit code is returned."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n, 1))))
    env["DV3_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def dry_launch(args):
    """The multi-rank plumbing without the HIP path: N ranks rendezvous (RCCL when every rank has its own GPU, else
    gloo on the host), exchange a flat gradient arena of the headline model's size through dist.BucketedAllReduce
    exactly as Trainer.optimizer_step does, check the sum, and rank 0 prints a contract-shaped JSON line."""
    from deepvoice3_pytorch_amd import dist as dv3dist
    import torch.distributed as tdist
    use_gpu = torch.cuda.is_available() and torch.cuda.device_count() >= max(args.gpus, 1)
    pg, rank, world, local_rank = dv3dist.init_from_env(backend=None if use_gpu else "gloo")
    dev = torch.device("cuda", local_rank) if use_gpu else torch.device("cpu")
    n_par = 48
    params = [torch.nn.Parameter(torch.zeros(256 * 1024 // 4 + 3 * i, device=dev)) for i in range(n_par)]   # ~12 MB
    from deepvoice3_pytorch_amd.train_step import FlatArena
    arena = FlatArena(params)
    comm = dv3dist.BucketedAllReduce(arena, pg, bucket_mb=2.0) if pg is not None else None
    t0 = time.perf_counter()
    for it in range(3):
        arena.grad.fill_(float(rank + 1))
        if comm is not None:
            comm.arm()
            for i in range(n_par - 1, -1, -1):      # gradients become final tail-first, as backward produces them
                comm._make_hook(i)(params[i])
            comm.finish()
        if use_gpu:
            torch.cuda.synchronize()
        want = world * (world + 1) / 2.0
        if float(arena.grad.min()) != want or float(arena.grad.max()) != want:
            raise RuntimeError("rank %d: all-reduced arena holds [%r, %r], expected %r"
                               % (rank, float(arena.grad.min()), float(arena.grad.max()), want))
    dt = (time.perf_counter() - t0) / 3
    backend = tdist.get_backend(pg) if pg is not None else "none"
    ranks = tdist.get_world_size(pg) if pg is not None else 1
    if pg is not None:
        tdist.barrier()
    if rank == 0:
        print(json.dumps(dict(metric="mel-frames/sec/node (train step, %s)" % args.preset, value=None, unit="mel-frames/s",
                              n_gpus=world, steps=0, warmup=0, ms_per_step=None, higher_is_better=True, scaling="weak",
                              vs_baseline=None, dtype=None, data="none (dry launch)", dry_launch=True,
                              rccl_ranks=ranks, backend=backend, device=dev.type,
                              allreduce_ms_per_arena=round(dt * 1e3, 3), arena_mb=round(arena.total * 4 / 2 ** 20, 1),
                              buckets=len(comm.buckets) if comm is not None else 0,
                              config=dict(workload="rank launch + bucketed gradient all-reduce only",
                                          parallelism="dp%d" % world))))
    if pg is not None:
        tdist.destroy_process_group()


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
    ap.add_argument("--gemm", default=None, choices=["f16x3", "bf16x3", "f32", "bf16"],
                    help="GEMM arithmetic (default: DV3_GEMM or f16x3)")
    ap.add_argument("--text-len", type=int, default=150)
    ap.add_argument("--frames", type=int, default=800)
    ap.add_argument("--no-graph", action="store_true", help="eager launches (default: probe both, keep the faster)")
    ap.add_argument("--graph", action="store_true", help="replay the whole step as one hipGraph")
    ap.add_argument("--settle", type=float, default=1.0,
                    help="seconds of untimed steps before --warmup (clock governor settle time; 0 = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip value_exact_f32 / configs / input_pipeline (headline + rooflines only)")
    ap.add_argument("--mode", default="train", choices=["train", "conv", "conv-ab", "synth"])
    ap.add_argument("--gl-iters", type=int, default=60, help="Griffin-Lim iterations (synth mode)")
    ap.add_argument("--force-group", action="store_true",
                    help="--gpus 1 only: run the headline step itself under a world-size-1 nccl (RCCL) process group "
                         "(gradient buckets armed); without it the same measurement appears under configs.ddp_world1")
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the N ranks, rendezvous and run the bucketed all-reduce only (no HIP compute; gloo "
                         "on a CPU-only box)")
    args = ap.parse_args()

    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        # no launcher around us: start the N ranks ourselves (the driver runs `python bench.py --gpus N`)
        if not args.dry_launch and torch.cuda.device_count() < args.gpus:
            sys.exit("bench.py --gpus %d: this node exposes %d GPU(s); one process per GPU is the only mode"
                     % (args.gpus, torch.cuda.device_count()))
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.dry_launch:
        return dry_launch(args)

    from deepvoice3_pytorch_amd import ops, dist as dv3dist
    if args.gemm:
        ops.set_gemm_precision(args.gemm)
    gemm = ops.gemm_precision()
    pg, rank, world, local_rank = dv3dist.init_from_env()
    if args.force_group and pg is None and args.gpus == 1:
        torch.cuda.set_device(0)
        _world1_group()
        import torch.distributed as tdist
        pg = tdist.group.WORLD
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    if args.mode == "synth":
        s = synth_run(dev, args.batch, reps=max(1, args.steps // 10), warm=max(1, args.warmup // 3),
                      gl_iters=args.gl_iters, step_graph=not args.no_graph)
        s.update(metric="real-time factor (synthesis: AR decode + converter + Griffin-Lim, %d concurrent utterances)"
                 % args.batch, n_gpus=1, steps=s["reps"], warmup=max(1, args.warmup // 3), scaling="weak",
                 vs_baseline=None, data="synthetic text ids, random-init weights")
        print(json.dumps(s))
        return
    if args.mode == "conv-ab":     # A/B of kernel variants / tiles at the north-star shape
        for hint in (0, 21, 22, 1, 2, 11, 12):
            for dil in (1, 27):
                rf = conv_roofline(dev, iters=10, tile_hint=hint, dil=dil)
                print("tile_hint=%2d dil=%2d  %8.1f us  %6.1f TFLOP/s  frac %.3f" % (hint, dil, rf["us_per_launch"], rf["achieved"], rf["frac"]))
        return
    if args.mode == "conv":
        rf = conv_roofline(dev, iters=max(args.steps, 10))
        print(json.dumps(dict(metric="conv1dglu_fwd_tflops", value=rf["achieved"], unit="TFLOP/s", n_gpus=1,
                              steps=args.steps, warmup=5, ms_per_step=rf["us_per_launch"] / 1e3,
                              higher_is_better=True, scaling="weak", vs_baseline=None, dtype=gemm,
                              data="synthetic", config=dict(workload="Conv1dGLU fwd B=64 C=256 T=1024 k=3"),
                              roofline=rf, roofline_wgrad=wgrad_roofline(dev))))
        return

    run = TrainRun(dev, pg, rank, world, args.preset, gemm, args.batch, args.text_len, args.frames, launch_mode(args, world))
    m = run.measure(args.steps, args.warmup, settle_s=args.settle)
    m_eager = None
    if run.use_graph and not args.no_extras:
        # the same step launched eagerly: host enqueue time and the all-reduce time left exposed are only observable there
        run.use_graph = False
        try:
            m_eager = run.measure(max(5, args.steps // 4), 3)
        finally:
            run.use_graph = True
    out = None
    if rank == 0:
        out = dict(metric="mel-frames/sec/node (train step, %s)" % args.preset, value=m["value"],
                   unit="mel-frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=m["ms_per_step"], higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype=gemm, dtype_note=dtype_note(gemm),
                   data="synthetic (fixed-shape LJSpeech-like: Tt=%d, %d frames/item; random-init weights)"
                   % (args.text_len, args.frames),
                   config=dict(workload="builder=%s preset=%s train step "
                                        "(fwd+losses+bwd+clip+Adam), synthetic LJSpeech-shaped batches" % (run.bname, args.preset),
                               per_gpu_batch=args.batch, global_batch=args.batch * world, text_len=args.text_len,
                               frames_per_item=args.frames, parallelism="dp%d" % world,
                               hipgraph=bool(run.use_graph), graph_form=run.graph_form(), launch_probe=run.launch_probe, gemm=gemm,
                               final_loss=m["final_loss"],
                               host_enqueue_ms_per_step=m["host_enqueue_ms_per_step"],
                               host_loop_ms_per_step=m["host_loop_ms_per_step"],
                               host_enqueue_note="time to issue one step into an idle queue (synchronize, then time the "
                                                 "issue); host_loop = host time per step inside the timed loop, which "
                                                 "includes the runtime's back-pressure once the host runs ahead",
                               launch_bound=bool(m["host_enqueue_ms_per_step"] > 0.97 * m["ms_per_step"]),
                               settle_s=args.settle),
                   step_flop_frac=m["step_flop_frac"])
        if pg is not None:
            import torch.distributed as tdist
            out["rccl_ranks"] = tdist.get_world_size(pg)
            out["backend"] = tdist.get_backend(pg)
            out["allreduce_exposed_ms"] = m["allreduce_exposed_ms"]
            out["gradient_buckets"] = len(run.trainer.comm.buckets)
        if run.graph_error:
            out["config"]["hipgraph_error"] = run.graph_error
        if m["power"]:
            out["avg_sclk_mhz"] = m["power"].get("avg_sclk_mhz")
            out["avg_power_w"] = m["power"].get("avg_power_w")
            out["power"] = m["power"]
        if m_eager is not None:
            out["eager"] = dict(value=m_eager["value"], ms_per_step=m_eager["ms_per_step"], steps=m_eager["steps"],
                                host_enqueue_ms_per_step=m_eager["host_enqueue_ms_per_step"],
                                launch_bound=bool(m_eager["host_enqueue_ms_per_step"] > 0.97 * m_eager["ms_per_step"]),
                                allreduce_exposed_ms=m_eager["allreduce_exposed_ms"],
                                note="the same step with per-kernel launches through Python + ctypes (--no-graph)")
    extras = not args.no_extras
    # ---- the same step fed through the input pipeline (sampler -> pinned staging -> side-stream H2D + device collate)
    if extras:
        try:
            from deepvoice3_pytorch_amd import data as dv3data
            rng = np.random.RandomState(99 + rank)
            n_items = args.batch * world * 4
            ds = dv3data.ListDataset(_bench_items(rng, min(n_items, 4 * args.batch), args.text_len, args.frames, run.hp),
                                     repeat=world)
            sampler = dv3data.LengthBucketedSampler(ds.frame_lengths, args.batch, rank=rank, world=world, seed=0)
            feed = dv3data.Prefetcher(ds, sampler, dev, outputs_per_step=1, downsample_step=4, depth=2, workers=4,
                                      loop=True, beside=[st for st in (torch.cuda.current_stream(), run.trainer.side_stream)
                                                         if st is not None])
            try:
                mf = run.measure(max(10, args.steps // 2), 5, feed=iter(feed))
            finally:
                feed.close()
            if rank == 0:
                out["input_pipeline"] = dict(value=mf["value"], ms_per_step=mf["ms_per_step"], steps=mf["steps"],
                                             note="every step consumes a fresh batch from data.Prefetcher: rank-sharded "
                                                  "length-bucketed sampler, ragged items copied into pinned staging by "
                                                  "worker threads, H2D + dv3_ragged_pad_rows on a side stream, double buffered")
        except Exception as e:
            if rank == 0:
                out["input_pipeline"] = dict(error="%s: %s" % (type(e).__name__, e))
    if rank == 0:
        # the boundary can also be handed host buffers (train.py:655-663 copies 8 tensors per step): time the
        # H2D of one pinned batch and report the rate with that copy serialised in front of every step
        bt, spk = run.bt, run.spk
        pinned = {k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in bt.items()}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            run.train_step.Batch.from_collate(pinned["text"], pinned["input_lengths"], pinned["mel"], pinned["y"],
                                              pinned["text_positions"], pinned["frame_positions"], pinned["done"],
                                              pinned["target_lengths"], spk, downsample_step=4, device=dev)
            torch.cuda.synchronize()
        h2d_ms = (time.perf_counter() - t0) / 3 * 1e3
        out["host_buffers"] = dict(h2d_ms_per_batch=round(h2d_ms, 3),
                                   value_with_serial_h2d=round(m["frames_per_step"] / world / ((m["ms_per_step"] + h2d_ms) * 1e-3) * world, 1),
                                   note="pinned host batch -> HBM copied synchronously before each step; never `value`")
    run.close()
    del run

    if extras and args.preset == "deepvoice3_ljspeech":
        if gemm != "f32":
            e = side_config(dev, pg, rank, world, args.preset, "f32", args, 5, 2)
            if rank == 0:
                out["value_exact_f32"] = e if "error" in e else dict(
                    value=e["value"], ms_per_step=e["ms_per_step"], steps=5, dtype="f32", dtype_note=dtype_note("f32"),
                    step_flop_frac=e["step_flop_frac"])
        cfgs = {}
        cfgs["nyanko_bf16"] = side_config(dev, pg, rank, world, "nyanko_ljspeech", "bf16", args, 20, 8)
        cfgs["vctk_bf16"] = side_config(dev, pg, rank, world, "deepvoice3_vctk", "bf16", args, 20, 8)
        if world == 1:
            cfgs["nyanko_" + gemm] = side_config(dev, pg, rank, world, "nyanko_ljspeech", gemm, args, 20, 8)
            try:
                s = synth_run(dev, 64, reps=2, warm=1, gl_iters=args.gl_iters)
                s["metric"] = "real-time factor (synthesis: AR decode + converter + Griffin-Lim, 64 concurrent utterances)"
                cfgs["synth_rtf"] = s
            except Exception as e:
                cfgs["synth_rtf"] = dict(error="%s: %s" % (type(e).__name__, e))
        if world == 1 and pg is None:
            # the reference's own operating points (presets/deepvoice3_ljspeech.json:48 batch_size 16; SURVEY 8d cfg2's
            # ragged LJSpeech-shaped lengths) and the data-parallel step armed on one GPU
            for key, kw in (("dv3lj_b16", dict(batch=16)), ("dv3lj_b64_ragged", dict(batch=args.batch, ragged=True)),
                            ("dv3lj_b16_ragged", dict(batch=16, ragged=True))):
                try:
                    cfgs[key] = side_config(dev, pg, rank, world, args.preset, gemm, args, 20, 8, **kw)
                except Exception as e:
                    cfgs[key] = dict(error="%s: %s" % (type(e).__name__, e))
            try:
                cfgs["dv3lj_b64_ragged_epoch"] = ragged_epoch_config(dev, args.preset, gemm, args)
            except Exception as e:
                cfgs["dv3lj_b64_ragged_epoch"] = dict(error="%s: %s" % (type(e).__name__, e))
            # the same epoch replayed from captured steps of a lattice of padded shapes (no host in the loop)
            for key, bsz, nb in (("dv3lj_b64_ragged_epoch_lattice", args.batch, 48), ("dv3lj_b16_ragged_epoch_lattice", 16, 64)):
                try:
                    cfgs[key] = ragged_lattice_config(dev, args.preset, gemm, args, bsz, n_batches=nb)
                except Exception as e:
                    import traceback
                    traceback.print_exc()
                    cfgs[key] = dict(error="%s: %s" % (type(e).__name__, e))
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
            try:
                # what one GPU can measure about the 8-GPU step's collective: a ring stand-in beside backward
                ds = dict(note="dist.RingStandin (csrc/standin.hip) in the communicator's place: 16 persistent workgroups x 256 "
                               "threads streaming 2*(7/8) of every gradient bucket at an assumed all-reduce busbw of 150 GB/s, "
                               "issued / awaited where the bucket all-reduces are; nothing is reduced (one GPU). "
                               "predicted_weak_scaling_efficiency = step without a group / step beside the stand-in; "
                               "sweeps of channels / busbw / bucket size: profiles/r06_collective_standin.txt")
                ds["dv3lj_" + gemm] = ddp_standin_config(dev, args.preset, gemm, args.batch, args, m["ms_per_step"])
                # (per-GPU batch 16: 8 MB buckets throughout -- measured x1.056 against x1.090 with the 25 MB default,
                #  profiles/r06_collective_standin.txt: a 25 MB bucket's wire time is 5 % of so short a step)
                ds["dv3lj_b16"] = ddp_standin_config(dev, args.preset, gemm, 16, args, cfgs.get("dv3lj_b16", {}).get("ms_per_step"),
                                                     bucket_mb=8.0)
                ds["nyanko_bf16"] = ddp_standin_config(dev, "nyanko_ljspeech", "bf16", args.batch, args, cfgs["nyanko_bf16"].get("ms_per_step"))
                ds["vctk_bf16"] = ddp_standin_config(dev, "deepvoice3_vctk", "bf16", args.batch, args, cfgs["vctk_bf16"].get("ms_per_step"))
                cfgs["ddp_standin"] = ds
            except Exception as e:
                cfgs["ddp_standin"] = dict(error="%s: %s" % (type(e).__name__, e))
            try:
                made = _world1_group()
                import torch.distributed as tdist
                d1 = dict(backend=tdist.get_backend(), rccl_ranks=tdist.get_world_size(),
                          note="world-size-1 nccl group in the bench process: gradient buckets armed, all-reduces on the "
                               "collective stream; eager = per-kernel launches on three streams, hipgraph = the "
                               "segmented replay, the bucket all-reduces issued from the host between segment launches "
                               "(nothing of the process group inside a capture)")
                d1["dv3lj_" + gemm] = ddp_world1_config(dev, args.preset, gemm, args, m["ms_per_step"])
                d1["nyanko_bf16"] = ddp_world1_config(dev, "nyanko_ljspeech", "bf16", args, cfgs["nyanko_bf16"].get("ms_per_step"))
                d1["vctk_bf16"] = ddp_world1_config(dev, "deepvoice3_vctk", "bf16", args, cfgs["vctk_bf16"].get("ms_per_step"))
                cfgs["ddp_world1"] = d1
                if made:
                    tdist.destroy_process_group()
            except Exception as e:
                cfgs["ddp_world1"] = dict(error="%s: %s" % (type(e).__name__, e))
        if rank == 0:
            out["configs"] = cfgs
    if rank != 0:
        _shutdown(pg)
        return
    if not args.no_roofline:
        rmode = "f16x3" if gemm == "bf16" else gemm
        out["roofline"] = conv_roofline(dev, mode=rmode)
        out["roofline_wgrad"] = wgrad_roofline(dev, mode=rmode)
        if gemm != "f32":      # the exact-fp32 kernel beside it, for the record
            rf = conv_roofline(dev, mode="f32", iters=30)
            out["roofline_exact_f32"] = dict(kernel=rf["kernel"], achieved=rf["achieved"], peak=rf["peak"],
                                             frac=rf["frac"], us_per_launch=rf["us_per_launch"])
        from deepvoice3_pytorch_amd import ops as _ops
        if _ops.bf16_storage:   # the same layer as the bf16 configs run it: single-term bf16 on c8 tensors
            rc = conv_roofline(dev, c8=True)
            out["roofline_bf16_c8"] = {k: rc[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                           "us_per_launch", "alg_flops", "alg_bytes", "hbm_gbs",
                                                           "hbm_frac", "variant")}
    if not args.no_cpu_baseline and world == 1 and args.preset == "deepvoice3_ljspeech":
        out["cpu_baseline"] = cpu_baseline(args.batch, args.text_len, args.frames)
    print(json.dumps(out), flush=True)
    _shutdown(pg)


def _shutdown(pg):
    """leave the job together: rank 0 still times the single-GPU rooflines after the other ranks are done, and a rank
    that exits while a peer's communicator is alive makes RCCL's watchdog noisy"""
    if pg is None:
        return
    import torch.distributed as tdist
    try:
        tdist.barrier()
        tdist.destroy_process_group()
    except Exception as e:       # the measurement is printed already: never turn a teardown problem into a failed run
        print("process group teardown: %s: %s" % (type(e).__name__, e), file=sys.stderr)


if __name__ == "__main__":
    main()
