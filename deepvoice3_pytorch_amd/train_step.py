# coding: utf-8
"""One optimisation step of the reference's train.train() (train.py:604-785) on the HIP path.

What the reference does on the host per step -- numpy guided-attention masks (train.py:594-601),
sequence masks, ~12 .item() syncs, per-tensor Adam -- is done here on the device:
    model forward (fused tap-GEMM layers)
    spec_loss (mel) + spec_loss (linear) + BCE(done) + guided attention loss, each ONE fused
        value+gradient kernel (csrc/loss.hip)
    backward (dgrad / wgrad tap-GEMMs, weight-norm backward)
    [data parallel: RCCL all-reduce of the flat gradient arena in reverse-layer buckets,
     launched on a side stream as soon as each bucket's gradients are final]
    global-norm clip + Adam over ONE flat parameter arena (csrc/optim.hip)
Scalars stay on the device; nothing in step() synchronises with the host.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import ops


class TrainConfig(object):
    """The hparams train.train() reads (hparams.py:96-121; presets/*.json)."""

    def __init__(self, outputs_per_step=1, downsample_step=4, masked_loss_weight=0.5,
                 binary_divergence_weight=0.1, priority_freq_weight=0.0, priority_freq=3000, sample_rate=22050,
                 use_guided_attention=True,
                 guided_attention_sigma=0.2, clip_thresh=0.1, adam_beta1=0.5, adam_beta2=0.9,
                 adam_eps=1e-6, weight_decay=0.0, initial_learning_rate=5e-4,
                 lr_schedule="noam_learning_rate_decay", lr_schedule_kwargs=None, max_positions=512,
                 range_check_every=0, amsgrad=False):
        self.outputs_per_step = outputs_per_step
        self.downsample_step = downsample_step
        self.masked_loss_weight = masked_loss_weight
        self.binary_divergence_weight = binary_divergence_weight
        # train.py:562-569,718-722: extra L1 weight on the bins below priority_freq (0 in every preset)
        self.priority_freq_weight, self.priority_freq, self.sample_rate = priority_freq_weight, priority_freq, sample_rate
        self.use_guided_attention = use_guided_attention
        self.guided_attention_sigma = guided_attention_sigma
        self.clip_thresh = clip_thresh
        self.adam_beta1, self.adam_beta2, self.adam_eps = adam_beta1, adam_beta2, adam_eps
        self.weight_decay = weight_decay
        # hparams.py:103 / train.py:975-979 hand `amsgrad` to optim.Adam (False in every preset).  The fused clip + Adam
        # launch keeps no running maximum of the second moment: accepted so that the reference's hparams can be forwarded
        # unchanged, refused explicitly when set
        if amsgrad:
            raise ValueError("amsgrad=True is not supported by the fused clip + Adam step (every reference preset uses "
                             "amsgrad=False, hparams.py:103)")
        self.amsgrad = False
        self.initial_learning_rate = initial_learning_rate
        self.lr_schedule = lr_schedule
        self.lr_schedule_kwargs = lr_schedule_kwargs or {}
        self.max_positions = max_positions
        # f16x3 range guard: every N steps read the device counter (one host sync) and move the run to the
        # bf16x3 mode when operands left the fp16 range; 0 = never read it on the host (the counter is still in
        # every step's scalars as a device tensor, "f16_range_events")
        self.range_check_every = range_check_every


def noam_learning_rate_decay(init_lr, global_step, warmup_steps=4000):
    """lrschedule.py:5-11."""
    warmup_steps = float(warmup_steps)
    step = global_step + 1.
    return init_lr * warmup_steps ** 0.5 * min(step * warmup_steps ** -1.5, step ** -0.5)


def step_learning_rate_decay(init_lr, global_step, anneal_rate=0.98, anneal_interval=30000):
    return init_lr * anneal_rate ** (global_step // anneal_interval)


def cyclic_cosine_annealing(init_lr, global_step, T, M):
    TdivM = T // M
    return init_lr / 2.0 * (math.cos(math.pi * ((global_step - 1) % TdivM) / TdivM) + 1.0)


_SCHEDULES = dict(noam_learning_rate_decay=noam_learning_rate_decay,
                  step_learning_rate_decay=step_learning_rate_decay,
                  cyclic_cosine_annealing=cyclic_cosine_annealing)


class Batch(object):
    """Device-resident training batch in the conventions of train.collate_fn (train.py:293-360):
    mel already time-downsampled (train.py:639-640), lengths as int32 device vectors."""

    def __init__(self, text, text_positions, frame_positions, mel, y, done, input_lengths_host,
                 target_lengths_host, speaker_ids, r, downsample_step, device):
        self.text, self.text_positions, self.frame_positions = text, text_positions, frame_positions
        self.mel, self.y, self.done, self.speaker_ids = mel, y, done, speaker_ids
        self.input_lengths_host = np.asarray(input_lengths_host)
        self.target_lengths_host = np.asarray(target_lengths_host)
        dl = self.target_lengths_host // r // downsample_step
        self.decoder_lengths_host = dl
        i32 = dict(dtype=torch.int32, device=device)
        self.input_lengths = torch.as_tensor(self.input_lengths_host.astype(np.int32), **i32)
        self.target_lengths = torch.as_tensor(self.target_lengths_host.astype(np.int32), **i32)
        self.decoder_lengths = torch.as_tensor(dl.astype(np.int32), **i32)
        # what the linear-domain mask is built from (train.py:669-677)
        self.linear_mask_lengths = self.target_lengths if downsample_step > 1 else self.decoder_lengths
        self.n_frames = int(self.target_lengths_host.sum())
        # ops.ValidLengths when the tensors are padded BEYOND the batch's own maxima (data.device_collate(lattice=),
        # data.pad_to_shape): the step then computes what it would on the batch padded to its own maxima
        self.valid = None

    @staticmethod
    def from_collate(x, input_lengths, mel, y, text_positions, frame_positions, done, target_lengths,
                     speaker_ids, downsample_step, device, r=1):
        if downsample_step > 1:
            mel = mel[:, 0::downsample_step, :].contiguous()
        f = lambda t: t.to(device, non_blocking=True) if t is not None else None
        return Batch(f(x.long()), f(text_positions.long()), f(frame_positions.long()), f(mel.float()),
                     f(y.float()), f(done.float()), input_lengths.numpy() if torch.is_tensor(input_lengths) else input_lengths,
                     target_lengths.numpy() if torch.is_tensor(target_lengths) else target_lengths,
                     f(speaker_ids), r, downsample_step, device)


class FlatArena(object):
    """All trainable parameters as views into one flat fp32 buffer; same for gradients and the
    Adam moments.  Order = reverse of first use in forward would be ideal for bucketing; we keep
    registration order and bucket from the tail (converter -> decoder -> encoder), which is the
    order backward produces gradients in."""

    def __init__(self, params):
        params = list(params)
        self.params = params
        dev = params[0].device
        sizes = [p.numel() for p in params]
        # 4-element alignment per tensor keeps every view 16-byte aligned
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 3) // 4 * 4
        self.offsets, self.sizes, self.total = offs, sizes, total
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o, n in zip(params, offs, sizes):
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
            p._dv3_grad_inplace = not os.environ.get("DV3_NO_INPLACE_GRAD")   # ops.ConvLayerFn accumulates straight into p.grad

    def rebind(self):
        """Every parameter's .grad must BE its slice of the gradient arena (clip/Adam and the all-reduce
        read the arena, nothing else).  model.zero_grad() / optimizer.zero_grad() default to
        set_to_none=True and autograd would then allocate fresh .grad tensors outside it: put the views
        back (the arena itself is zeroed by the caller).  A .grad that points elsewhere is an error."""
        for p, o, n in zip(self.params, self.offsets, self.sizes):
            g = p.grad
            if g is None:
                p.grad = self.grad[o:o + n].view(p.shape)
            elif g.data_ptr() != self.grad.data_ptr() + 4 * o:
                raise RuntimeError("a parameter's .grad no longer aliases the flat gradient arena "
                                   "(was it replaced by an optimizer or a manual assignment?)")


class Trainer(object):
    HYPER_SLOTS = 8

    def __init__(self, model, cfg, global_step=0, process_group=None, bucket_mb=25.0, last_bucket_mb=8.0,
                 train_seq2seq=True, train_postnet=True):
        """train_seq2seq / train_postnet: the reference's `--train-seq2seq-only` / `--train-postnet-only` runs
        (train.py:608-616, 684-731): only model.seq2seq (mel + done + attention losses) or only model.postnet on the mel
        TARGETS (linear loss) is run and updated.  The reference keeps ONE optimizer over every trainable parameter and
        lets Adam skip those without a gradient; here the arena holds the trained sub-module's parameters only (the others
        take no gradient, no moment decay and no update -- what skipping means), numbered as the reference's optimizer
        numbers them (`optimizer_order`), so checkpoints interoperate in both modes."""
        assert train_seq2seq or train_postnet
        self.train_seq2seq, self.train_postnet = bool(train_seq2seq), bool(train_postnet)
        self.model, self.cfg = model, cfg
        self.global_step = global_step
        self.adam_step = 0
        params = list(model.get_trainable_parameters())
        # The optimizer-facing order (checkpoint interop: torch.optim.Adam numbers its state by this order) ...
        self.optimizer_order = list(params)
        if not (self.train_seq2seq and self.train_postnet):
            sub = model.seq2seq if self.train_seq2seq else model.postnet
            inside = set(id(p) for p in sub.parameters())
            params = [p for p in params if id(p) in inside]
        # ... and the arena's.  Data parallel, multi-speaker: the speaker projections of the Conv1dGLU layers get their
        # gradients from ONE backward node per block (ops.SpeakerBiasBlockFn) that runs when the whole block's backward is
        # done -- left between their layers' weights they would hold every bucket of the block back until then.  They
        # go to the arena's tail as a group (one small bucket of their own, final when the encoder's backward is).
        self.late_group = []
        if process_group is not None and ops.fused_speaker_bias:
            late = set(id(p) for n, p in model.named_parameters() if ".speaker_proj." in "." + n)
            self.late_group = [p for p in params if id(p) in late]
            params = [p for p in params if id(p) not in late] + self.late_group
        self.arena = FlatArena(params)
        # parameters the optimizer does not own (the frozen position tables, the text embedding under
        # freeze_embedding; reference __init__.py:48-63) take no gradient at all: a .grad outside the arena
        # would never be zeroed and only cost backward work
        owned = set(id(p) for p in params)
        self._frozen = []                 # (parameter, its requires_grad before): restored by close()
        for p in model.parameters():
            if id(p) not in owned:
                self._frozen.append((p, p.requires_grad))
                p.requires_grad_(False)
        dev = self.arena.flat.device
        self.device = dev
        self._prepack = None
        self.hyper = torch.zeros(3, dtype=torch.float32, device=dev)
        # (lr, 1-b1^t, sqrt(1-b2^t)) travel through a RING of pinned host slots: step() never waits for the
        # GPU, so the host may run steps ahead of it; a single staging buffer would be overwritten by step
        # N+1 before step N's queued copy has executed.  A slot is reused only after the copy that read it
        # has completed (its event), i.e. the host can lead by at most HYPER_SLOTS - 1 steps.
        self._hyper_ring = [(torch.zeros(3, dtype=torch.float32).pin_memory() if dev.type == "cuda"
                             else torch.zeros(3)) for _ in range(self.HYPER_SLOTS)]
        self._hyper_events = [None] * self.HYPER_SLOTS
        self._hyper_slot = 0
        self.norm_partial = torch.empty(1024, dtype=torch.float32, device=dev)
        self.norm_out = torch.zeros(2, dtype=torch.float32, device=dev)
        # weight-norm backward of 8 layers per launch (ops.WnBwdBatch): opt-in (DV3_WN_BWD_BATCH=1).  Bit-identical, but
        # measured 2 % SLOWER in the step (15.62 vs 15.29 ms, nyanko bf16 11.79 vs 11.53): the per-layer launches
        # already overlap with the input-gradient chain on the side stream, a fat launch every 8 layers competes with it
        self.batch_wn_bwd = os.environ.get("DV3_WN_BWD_BATCH", "0") == "1"
        # second stream for the weight-gradient branch of backward (ops.SideStream); DV3_WGRAD_STREAM=0 keeps one stream
        # (a stream that shares no hardware queue with the step stream: ops.concurrent_stream)
        self.side_stream = None
        self.side_stream_beside = False
        if dev.type == "cuda" and os.environ.get("DV3_WGRAD_STREAM", "1") not in ("0", ""):
            with torch.cuda.device(dev):
                # DV3_SIDE_PRIORITY=low: the weight-gradient stream at the device's least stream priority (ops.new_stream)
                self.side_stream = ops.concurrent_stream([torch.cuda.current_stream()], role="weight-gradient",
                                                         priority=os.environ.get("DV3_SIDE_PRIORITY", "normal"))
                # did the probe SEE this stream run beside the step stream?  (GraphedTrainer.flag_sync needs that: a wait
                # kernel on a queue it shares with its signal would sit out its time-out)
                rec = ops.stream_probe_log[-1] if ops.stream_probe_log else {}
                self.side_stream_beside = bool(rec.get("probed") and rec.get("found"))
        self.pg = process_group
        self.world = 1
        self.comm = None
        if process_group is not None:
            from . import dist as _dist
            # (a dist.RingStandin measures a ring's footprint on one GPU: the bucket schedule is the data-parallel step's,
            # the arithmetic the single-GPU step's)
            self.world = 1 if getattr(process_group, "is_standin", False) else torch.distributed.get_world_size(process_group)
            # the speaker table receives a gradient from every layer of every module: a bucket of its own
            shared = set(id(p) for n, p in model.named_parameters() if n.split(".")[-2:-1] == ["embed_speakers"])
            isolate = [i for i, p in enumerate(self.arena.params) if id(p) in shared]
            group_start = len(self.arena.params) - len(self.late_group) if self.late_group else None
            # the all-reduces are issued (asynchronously) from the weight-gradient stream itself: no collective stream of
            # the trainer's own to share a hardware queue with (dist.BucketedAllReduce); DV3_COLLECTIVE_STREAM=own: the
            # round-4 form (a probed fourth stream), kept for A/B runs
            own = os.environ.get("DV3_COLLECTIVE_STREAM", "") == "own" or self.side_stream is None
            self.comm = _dist.BucketedAllReduce(self.arena, process_group, bucket_mb, last_bucket_mb, isolate=isolate, boundaries=() if group_start is None else (group_start,),
                                                beside=[st for st in (torch.cuda.current_stream() if dev.type == "cuda" else None,
                                                                      self.side_stream) if st is not None],
                                                issue_stream=None if own else self.side_stream)

    def checkpoint_module(self):
        """what train.save_checkpoint stores (train.py:788-797): the model, or the trained sub-module in a split mode"""
        if self.train_seq2seq and self.train_postnet:
            return self.model
        return self.model.seq2seq if self.train_seq2seq else self.model.postnet

    def checkpoint_suffix(self):
        if self.train_seq2seq and self.train_postnet:
            return ""
        return "_seq2seq" if self.train_seq2seq else "_postnet"

    def close(self):
        """Detach the gradient-exchange hooks (call before building another Trainer on the same model)."""
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        for p, flag in getattr(self, "_frozen", []):      # hand the un-owned parameters back as they were
            p.requires_grad_(flag)
        self._frozen = []

    # ------------------------------------------------------------------------------------
    def current_lr(self):
        c = self.cfg
        if c.lr_schedule is None:
            return c.initial_learning_rate
        return _SCHEDULES[c.lr_schedule](c.initial_learning_rate, self.global_step, **c.lr_schedule_kwargs)

    def _set_hyper(self):
        c = self.cfg
        self.adam_step += 1
        t = self.adam_step
        i = self._hyper_slot
        self._hyper_slot = (i + 1) % self.HYPER_SLOTS
        if self._hyper_events[i] is not None:
            self._hyper_events[i].synchronize()       # the copy that read this slot has run
        h = self._hyper_ring[i]
        h[0] = float(self.current_lr())
        h[1] = 1.0 - c.adam_beta1 ** t
        h[2] = math.sqrt(1.0 - c.adam_beta2 ** t)
        self.hyper.copy_(h, non_blocking=True)
        if self.device.type == "cuda":
            ev = self._hyper_events[i] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._hyper_events[i] = ev

    def check_lengths(self, batch):
        """train.py:646-652."""
        max_seq_len = max(batch.input_lengths_host.max(), batch.decoder_lengths_host.max())
        if max_seq_len >= self.cfg.max_positions:
            raise RuntimeError(
                """max_seq_len ({}) >= max_posision ({})
Input text or decoder targget length exceeded the maximum length.
Please set a larger value for ``max_position`` in hyper parameters.""".format(max_seq_len, self.cfg.max_positions))

    def _prepack_all(self):
        """weight norm + both split operand images of every Conv1d / Linear layer in two launches"""
        if ops.gemm_precision() == "f32" or self.device.type != "cuda":
            return None
        if self._prepack is None or self._prepack.mode != ops.gemm_precision():
            from . import conv as _conv, modules as _modules
            gated = {id(m.conv): m.conv.out_channels // 2 for m in self.model.modules()
                     if isinstance(m, _modules._GatedConv)}
            layers = []
            for m in self.model.modules():
                if isinstance(m, _conv._WNLayer) and not m.transposed:
                    v, g = m.wn_params()
                    if v.is_contiguous() and (g is None or g.is_contiguous()):
                        layers.append((v, g, gated.get(id(m), 0)))
            self._prepack = ops.Prepack(layers) if layers else None
        if self._prepack is not None:
            self._prepack.run()
        return self._prepack

    # ------------------------------------------------------------------------------------
    def forward_backward(self, batch):
        """model forward + losses + backward.  Returns the scalars as device tensors."""
        c = self.cfg
        r = c.outputs_per_step
        self.arena.rebind()
        for p in self.arena.params:      # in-place gradient bookkeeping of ops.ConvLayerFn (uses this step)
            p._dv3_pending = 0
        self.model.train()
        ops.prepacked = self._prepack_all()
        ops.mask_plan.begin_step(self.device)      # the step's dropout masks in one launch once its list of sites repeats
        s2s, pn = self.train_seq2seq, self.train_postnet
        # a batch padded beyond its own maxima (data.pad_to_shape: the lattice shapes of LatticeReplay) carries those maxima
        # as device scalars; the model's non-causal stacks, its attention and the loss means below read them
        vl = ops.valid = getattr(batch, "valid", None)
        v_mel, v_lin, v_dec, v_in = (vl.tv[2:3], vl.tv[3:4], vl.tv[1:2], vl.tv[0:1]) if vl is not None else (None,) * 4
        self._valid_args = (v_mel, v_lin, v_dec, v_in)
        try:
            if s2s and pn:
                mel_out, lin_out, attn, done_hat = self.model(
                    batch.text, batch.mel, speaker_ids=batch.speaker_ids, text_positions=batch.text_positions,
                    frame_positions=batch.frame_positions, input_lengths=batch.input_lengths)
            elif s2s:                                    # train.py:689-697
                assert batch.speaker_ids is None
                mel_out, attn, done_hat, _ = self.model.seq2seq(
                    batch.text, batch.mel, text_positions=batch.text_positions,
                    frame_positions=batch.frame_positions, input_lengths=batch.input_lengths)
                mel_out = mel_out.reshape(batch.mel.size(0), -1, batch.mel.size(-1))
                lin_out = None
            else:                                        # train.py:698-701: the post-net on the mel TARGETS
                assert batch.speaker_ids is None
                lin_out = self.model.postnet(batch.mel)
                mel_out = attn = done_hat = None
        finally:
            ops.prepacked = None
            ops.valid = None
            ops.mask_plan.end_step()
        if not (s2s and pn):
            return self._split_losses_backward(batch, mel_out, lin_out, attn, done_hat)
        wm, w = c.masked_loss_weight, c.binary_divergence_weight
        lin_len = batch.linear_mask_lengths if wm > 0 else None
        direct = c.priority_freq_weight <= 0 and self.device.type == "cuda"
        if direct:
            # The fused loss kernels write value AND gradient in one pass and every term enters the total with weight 1
            # (train.py:728-740): the gradients go straight to autograd.backward below -- no select / mul / ones / add
            # nodes between the loss terms and the model outputs (round 3: 16 torch launches per step, two of them
            # 100 MB multiplications by 1.0).
            m4, g_mel = ops.spec_loss_with_grad(mel_out, batch.mel, batch.decoder_lengths if wm > 0 else None, r, wm, w,
                                                t_valid=v_mel)
            l4, g_lin = ops.spec_loss_with_grad(lin_out, batch.y, lin_len, r, wm, w, t_valid=v_lin)
            done_loss, g_done = ops.bce_loss_with_grad(done_hat, batch.done, t_valid=v_dec)
            lin_loss = l4[2]
            roots, grads = [mel_out, lin_out, done_hat], [g_mel, g_lin, g_done]
            attn_loss = None
            if c.use_guided_attention:
                attn_loss, g_attn = ops.guided_attention_loss_with_grad(attn, batch.input_lengths, batch.decoder_lengths,
                                                                         c.guided_attention_sigma, v_dec, v_in)
                roots.append(attn)
                grads.append(g_attn)
            loss = ops.sum_scalars(m4[2:3], l4[2:3], done_loss, attn_loss)
        else:
            m4 = ops.spec_loss(mel_out, batch.mel, batch.decoder_lengths if wm > 0 else None, r, wm, w, v_mel)
            l4 = ops.spec_loss(lin_out, batch.y, lin_len, r, wm, w, v_lin)
            lin_loss = l4[2]
            if c.priority_freq_weight > 0:
                # l1 := (1 - pw) * l1 + pw * l1(first n bins)   (train.py:562-569): two more passes of the fused loss
                # kernel with the binary-divergence weight 0, whose third output IS the (masked) L1 and carries gradient
                n_pri = int(c.priority_freq / (c.sample_rate * 0.5) * lin_out.size(-1))
                l1_all = ops.spec_loss(lin_out, batch.y, lin_len, r, wm, 0.0, v_lin)[2]
                l1_pri = ops.spec_loss(lin_out[:, :, :n_pri], batch.y[:, :, :n_pri], lin_len, r, wm, 0.0, v_lin)[2]
                lin_loss = lin_loss + (1.0 - w) * c.priority_freq_weight * (l1_pri - l1_all)
            done_loss = ops.bce_loss(done_hat, batch.done, v_dec)
            loss = m4[2] + lin_loss + done_loss[0]
            attn_loss = None
            if c.use_guided_attention:
                attn_loss = ops.guided_attention_loss(attn, batch.input_lengths, batch.decoder_lengths,
                                                      c.guided_attention_sigma, v_dec, v_in)
                loss = loss + attn_loss[0]
        scal = dict(mel_l1_loss=m4[0], mel_binary_div_loss=m4[1], mel_loss=m4[2], linear_l1_loss=l4[0],
                    linear_binary_div_loss=l4[1], linear_loss=lin_loss, done_loss=done_loss[0])
        if attn_loss is not None:
            scal["attn_loss"] = attn_loss[0]
        scal["loss"] = loss[0] if direct else loss
        if self.comm is not None:
            self.comm.arm()
        ops.SideStream.stream = self.side_stream
        ops.SideStream.main = torch.cuda.current_stream() if self.side_stream is not None else None
        ops.SideStream.capturing = self.side_stream is not None and torch.cuda.is_current_stream_capturing()
        # weight-norm backward of 8 layers per launch (ops.WnBwdBatch) unless gradient-ready hooks want every
        # parameter's gradient as early as possible (data parallel buckets)
        ops.WnBwdBatch.active = self.batch_wn_bwd and not ops.grad_ready_hooks
        try:
            if direct:
                torch.autograd.backward(roots, grads)
            else:
                loss.backward()
            if ops.WnBwdBatch.active:       # the layers still queued, on the stream their weight gradients ran on
                if self.side_stream is not None:
                    with ops.SideStream._section:
                        ops.WnBwdBatch.flush()
                else:
                    ops.WnBwdBatch.flush()
        finally:
            ops.WnBwdBatch.active = False
            ops.WnBwdBatch.discard()
            ops.SideStream.join()          # the step stream waits for the weight-gradient branch; its operands may go
            ops.SideStream.stream = ops.SideStream.main = None
        return {k: v.detach() for k, v in scal.items()}

    def _split_losses_backward(self, batch, mel_out, lin_out, attn, done_hat):
        """losses + backward of the seq2seq-only / postnet-only steps (train.py:704-740): the joint step's terms, those
        of the part that ran"""
        c = self.cfg
        r, wm, w = c.outputs_per_step, c.masked_loss_weight, c.binary_divergence_weight
        v_mel, v_lin, v_dec, v_in = self._valid_args
        scal, terms = {}, []
        if self.train_seq2seq:
            m4 = ops.spec_loss(mel_out, batch.mel, batch.decoder_lengths if wm > 0 else None, r, wm, w, v_mel)
            done_loss = ops.bce_loss(done_hat, batch.done, v_dec)
            scal.update(mel_l1_loss=m4[0], mel_binary_div_loss=m4[1], mel_loss=m4[2], done_loss=done_loss[0])
            terms += [m4[2], done_loss[0]]
            if c.use_guided_attention:
                attn_loss = ops.guided_attention_loss(attn, batch.input_lengths, batch.decoder_lengths,
                                                      c.guided_attention_sigma, v_dec, v_in)
                scal["attn_loss"] = attn_loss[0]
                terms.append(attn_loss[0])
        else:
            lin_len = batch.linear_mask_lengths if wm > 0 else None
            l4 = ops.spec_loss(lin_out, batch.y, lin_len, r, wm, w, v_lin)
            lin_loss = l4[2]
            if c.priority_freq_weight > 0:
                n_pri = int(c.priority_freq / (c.sample_rate * 0.5) * lin_out.size(-1))
                l1_all = ops.spec_loss(lin_out, batch.y, lin_len, r, wm, 0.0, v_lin)[2]
                l1_pri = ops.spec_loss(lin_out[:, :, :n_pri], batch.y[:, :, :n_pri], lin_len, r, wm, 0.0, v_lin)[2]
                lin_loss = lin_loss + (1.0 - w) * c.priority_freq_weight * (l1_pri - l1_all)
            scal.update(linear_l1_loss=l4[0], linear_binary_div_loss=l4[1], linear_loss=lin_loss)
            terms.append(lin_loss)
        loss = terms[0]
        for t in terms[1:]:
            loss = loss + t
        scal["loss"] = loss
        if self.comm is not None:
            self.comm.arm()
        ops.SideStream.stream = self.side_stream
        ops.SideStream.main = torch.cuda.current_stream() if self.side_stream is not None else None
        ops.SideStream.capturing = self.side_stream is not None and torch.cuda.is_current_stream_capturing()
        try:
            loss.backward()
        finally:
            ops.SideStream.join()
            ops.SideStream.stream = ops.SideStream.main = None
        return {k: v.detach() for k, v in scal.items()}

    def optimizer_step(self, reduce=True):
        """reduce=False: the caller has already issued and joined the gradient all-reduces (GraphedTrainer's segmented
        replay issues them from the host between segment launches)"""
        c, a = self.cfg, self.arena
        if self.comm is not None and reduce:
            self.comm.finish()           # all buckets reduced (sum); 1/world folded into grad_prescale
        prescale = 1.0 / self.world
        if c.clip_thresh > 0:
            ops.grad_sqnorm(a.grad, self.norm_partial, self.norm_out)   # norm of the SUMMED gradient
        ops.clip_adam(a.flat, a.grad, a.exp_avg, a.exp_avg_sq, self.norm_out if c.clip_thresh > 0 else None,
                      c.clip_thresh, self.hyper, c.adam_beta1, c.adam_beta2, c.adam_eps, c.weight_decay,
                      prescale)

    def step(self, batch):
        """One full optimisation step; returns device scalars (loss terms, grad_norm, lr)."""
        self.check_lengths(batch)        # host-side numpy on the batch's length vectors: no device sync
        self._set_hyper()
        self._zero_grad()
        scal = self.forward_backward(batch)
        n = self.cfg.range_check_every
        if n and (self.global_step + 1) % n == 0 and self.check_range():
            # operands left the fp16 range in THIS step's forward (or since the last check): its gradients may hold
            # Inf / NaN -- they must not reach clip / Adam.  check_range() has moved the run to bf16x3 (same decision
            # on every rank); the step is redone there before anything is applied.
            if self.comm is not None:
                self.comm.finish()       # the buckets of the discarded backward are in flight: join them first
            self._zero_grad()
            scal = self.forward_backward(batch)
        self.optimizer_step()
        scal["grad_norm"] = self._scalar(self.norm_out[0:1], 1.0 / self.world)
        scal["learning_rate"] = self._scalar(self.hyper[0:1])
        if ops.gemm_precision() == "f16x3" and self.device.type == "cuda":
            scal["f16_range_events"] = ops.f16_range_events_tensor(self.device)
        self.global_step += 1
        return scal

    def _zero_grad(self):
        """optimizer.zero_grad() (train.py:683) on the flat gradient arena: one hipMemsetAsync through the library"""
        if self.device.type == "cuda":
            ops.zero_(self.arena.grad)
        else:
            self.arena.grad.zero_()

    def _scalar(self, t1, alpha=1.0):
        """alpha * t1 (a one-element device tensor) as its own tensor, without a torch kernel; -> 0-dim"""
        if self.device.type == "cuda":
            return ops.scaled_copy(t1, alpha)[0]
        return (t1 * alpha)[0]

    def check_range(self):
        """f16x3 range guard (ops.f16_range_events): when forward operands left the fp16 range since the last check
        ON ANY RANK (MAX all-reduce: every rank takes the same decision and stays in the same GEMM mode), warn and
        continue in the bf16x3 mode (full exponent range; the weight images are re-packed on the next forward).  One
        host synchronisation.  Called by step() BEFORE the update is applied (every `range_check_every` steps), so a
        poisoned gradient is never stepped into the parameters.  A GraphedTrainer can not switch mode inside its
        captured graph: see GraphedTrainer.check_range.  -> number of events (max over ranks)"""
        if ops.gemm_precision() != "f16x3" or self.device.type != "cuda":
            return 0
        n = ops.f16_range_events(reset=True, device=self.device)
        if self.pg is not None and self.world > 1:
            t = torch.tensor([n], dtype=torch.int64, device=self.device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=self.pg)
            n = int(t.item())
        if n:
            import warnings
            warnings.warn("%d operand units left the fp16 range of the f16x3 GEMM mode (|activation| > 4094 or "
                          "|weight| > 255.9): continuing in the bf16x3 mode" % n)
            ops.set_gemm_precision("bf16x3")
        return n


_CAPTURE_STREAMS = {}


def _capture_stream(device, avoid=()):
    """ONE capture stream per device for every GraphedTrainer of the process: ops._sk_ws keeps a permanent
    67 MB stream-K workspace per (device, stream) -- a fresh torch.cuda.Stream() per capture cycled through torch's 32
    pool streams and pinned up to 2 GB, and a pool handle shared with an unrelated eager user would have shared the
    workspace's flags with a replaying graph.  Captures are sequential, replays run on the caller's stream."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _CAPTURE_STREAMS.get(key)
    if st is None:
        taken = set(a.cuda_stream for a in avoid if a is not None)     # (the first capture's trainer already has its streams)
        with torch.cuda.device(key):
            for _ in range(40):
                st = ops.new_stream()
                if st.cuda_stream not in taken:
                    break
        _CAPTURE_STREAMS[key] = st
        ops.reserved_stream_handles.add(st.cuda_stream)      # no later side / prefetch / collective stream gets this handle
    return st


class GraphedTrainer(object):
    """Whole-step hipGraph replay: forward + losses + backward + (bucketed all-reduce) + clip/Adam captured once per
    batch shape and replayed -- removes the per-kernel host launch cost (~330 launches/step: 5-16 ms of host time,
    box dependent, against a 12-16 ms step at the north-star batch).  Dropout masks still change every replay: the
    Philox seed offset lives on the device.  `step(batch)` copies a fresh batch of the captured shape into the static
    one first (device-to-device, on the step stream); batches of another shape raise -- bucket the input pipeline by
    shape (data.LengthBucketedSampler pads to the batch maximum) or run those eagerly.

    split_streams (default when the trainer has its second backward stream).  A single captured step replays with its
    two backward branches serialised (4-7 % slower than eager launches whenever the GPU is the bound:
    profiles/r03_side_stream_ab.txt), so the step is captured as SEGMENTS and the weight-gradient branch
    (ops.SideStream) is replayed on the real second stream:
        step-stream segment j   torch.cuda.CUDAGraph: (zero_grad, forward, losses for j = 0,) the input-gradient chain
                                of `chunk` layers
        side segment j          the side stream's own capture over the same stretch (include/dv3hip.h:
                                dv3_graph_side_begin / _end): weight-gradient GEMMs + weight-norm backward of those layers
        tail                    clip + Adam (under a process group: after the host-issued all-reduces, see below)
    Replay: for j: launch step segment j; record an ordinary event; the side stream waits for it; launch side segment j.
    Then the step stream waits for the side stream and the tail runs.  Every dependency is a host-issued
    hipEventRecord / hipStreamWaitEvent (event NODES between two graphs were tried first and read stale at the
    benchmark's sizes: DESIGN.md 3.7).  Measured with the first form, which has the same overlap
    (profiles/r04_three_graph_probe.txt): the eager step's GPU time at the replay's host cost -- 15.64 vs 16.44 ms (one
    graph) vs 15.62 (eager) for deepvoice3_ljspeech f16x3 B=64; 12.46 vs 13.30 vs 16.17 (eager, host bound on that box)
    for deepvoice3_vctk bf16.

    Data parallel.  With split_streams NOTHING of the process group is captured: while the segments are captured,
    dist.BucketedAllReduce notes which segment completes which bucket, and the replay issues those all-reduces from the
    host on the collective stream right after launching that segment (it waits for both streams' events of the segment)
    -- the same overlap with the rest of backward as the eager step, the same order on every rank (the capture is a
    function of the model), and the ordinary c10d call path.  The buckets no segment completed (parameters without a
    gradient) go before the tail, which the step stream enters after joining the side and the collective stream.
    (Round 4's first form captured the collectives into the tail graph: c10d's watchdog thread then queried an event
    last recorded in a capturing stream -- hipErrorCapturedEvent, process abort -- once in three runs at the
    benchmark's sizes.)  Without split_streams (single graph) the collectives are still captured with the step, in
    the thread-local capture mode."""

    def __init__(self, trainer, static_batch, warmup=3, split_streams=None, chunk=None, dry_warmup=False, pool=None):
        """dry_warmup: the warm-up passes run forward + backward only (no clip / Adam: parameters, moments and the step
        counter stay as they are) -- what a LatticeReplay wants when it meets a new padded shape in the middle of a run.
        pool: a torch.cuda.graph_pool_handle() shared with other GraphedTrainers of the same trainer that are never
        replayed concurrently (LatticeReplay: one step at a time, whatever its shape) -- their activations then share
        one set of blocks instead of holding a private set per captured shape."""
        self.t = trainer
        self.batch = static_batch
        self._pool = pool
        trainer.check_lengths(static_batch)
        dev = trainer.device
        if split_streams is None:
            split_streams = trainer.side_stream is not None and os.environ.get("DV3_SPLIT_GRAPH", "1") not in ("0", "")
        self.split = bool(split_streams) and trainer.side_stream is not None
        self.chunk = int(chunk or os.environ.get("DV3_SPLIT_CHUNK", "0"))      # 0: chosen after the warm-up steps
        self.cut_on_bucket = os.environ.get("DV3_CUT_ON_BUCKET", "1") not in ("0", "")
        self.tail_fine = int(os.environ.get("DV3_SPLIT_TAIL", "6"))     # the last N fork points end a segment each (0 = off)
        self.head_fine = int(os.environ.get("DV3_SPLIT_HEAD", "0"))     # ... and the first N (0 = off)
        self.n_forks = 0
        # ABI 43 (round 6, last part): backward as ONE graph per stream, its fork points ordered by a device flag
        # (include/dv3hip.h: dv3_flag_signal / dv3_flag_wait) instead of segment boundaries.  Not under a process group:
        # the host-issued collectives of the segmented form need the segments.  DV3_FLAG_SYNC=0: the segments.
        # By rule (profiles/r06i_flag_sync.txt): it wins where the segment boundaries are a visible share of the step --
        # fp32 activations below batch 48 (deepvoice3_ljspeech B = 16: -1.1 ... -1.8 %) -- is even at B = 64 (the boundaries'
        # 0.23 ms against 44 signal kernels on the critical queue) and LOSES with channel-blocked bf16 activations
        # (+2.4 ... +3.6 %: 58 signal kernels of ~3 us each on a 9 ms step's critical queue, and weight gradients that start
        # the moment their operands exist take compute units from the input-gradient chain).
        # OPT-IN (DV3_FLAG_SYNC=rule: by that rule; =1: always; default 0: the segments): HIP maps streams to hardware
        # queues dynamically, and one bench process showed a replay of this form at 11 ms against 6.2 ms -- two graphs whose
        # streams share a queue run one after the other, where the segments only lose their overlap.  bench.py probes it per
        # configuration (like eager against the replay) and keeps it where it is the faster form in that process.
        fs = os.environ.get("DV3_FLAG_SYNC", "0")
        small = int(static_batch.mel.size(0)) < 48
        self.flag_sync = (self.split and trainer.comm is None and
                          (fs == "1" or (fs not in ("0", "") and small and not ops.storage_c8() and
                                         getattr(trainer, "side_stream_beside", False))))
        self._flag = torch.zeros(4, dtype=torch.int64, device=dev) if self.flag_sync else None   # flag, epochs, err
        self._flag_checked = False
        self.flag_lag = int(os.environ.get("DV3_FLAG_LAG", "0"))
        # a signal kernel at every k-th fork point only (profiles/r06i_flag_sync.txt: k = 2 brings the bf16 presets to parity
        # and B = 64 to -0.5 %, at -1.3 % instead of -1.8 % for B = 16, which is what the rule serves: k = 1)
        self.flag_every = int(os.environ.get("DV3_FLAG_EVERY", "1"))
        self.seed_offset = torch.zeros(1, dtype=torch.int64, device=dev)
        self._prev_offset = ops.dropout_state.dev_offset        # restored by close()
        ops.dropout_state.dev_offset = self.seed_offset
        s = _capture_stream(dev, avoid=(trainer.side_stream, torch.cuda.current_stream()))   # (the warm-up steps' stream-K launches take their workspace per stream too)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):      # real optimisation steps (lr / bias corrections set first)
                if dry_warmup:           # ... or forward + backward only (_set_hyper would count an Adam step)
                    trainer._zero_grad()
                    trainer.forward_backward(self.batch)
                    continue
                trainer._set_hyper()
                self._body()
                trainer.global_step += 1
        torch.cuda.current_stream().wait_stream(s)
        site0 = ops.dropout_state.site
        self.segs, self._seg_events, self._join_event, self.graph2 = [], [], None, None
        self.seg_buckets, self.rest_buckets = [], []
        if self.chunk <= 0:
            # fork points per segment.  Measured (profiles/r04_split_chunk_ab.txt): flat between 4 and 10 -- 2 costs graph
            # launches, 20 and more lose the overlap (one segment = the single graph's time); the 46-fork
            # deepvoice3_ljspeech step is 0.5 % faster at 4 (15.20 vs 15.28 ms), the 58-fork nyanko step 0.9 % faster at
            # 10 (10.80 vs 10.90 ms), batch 16 the same at both
            # Round 6 re-scan (profiles/r06_split_chunk.txt): what separates the two was the storage mode, not the fork
            # count -- with channel-blocked bf16 activations (short kernels) 8 (10 for the 58-fork nyanko step) is the best or within noise of it
            # (deepvoice3_vctk bf16 -1.7 % against 4, nyanko bf16 8 ~ 10 < 6 < 16), with fp32 activations 4 is (nyanko f16x3
            # +1.7 % at 8, and it used to get 10 by its fork count)
            n = ops.SideStream.forks_last if warmup > 0 else 0
            small = int(static_batch.mel.size(0)) < 48          # short kernels again: B = 16 / 32 want 5-6 (-0.8 % / -0.5 %)
            self.chunk = (8 if 0 < n < 52 else 10) if ops.storage_c8() else (6 if small else 4)
        self.n_forks = ops.SideStream.forks_last if warmup > 0 else 0
        if not self.split:
            # a process group brings its watchdog thread: its event queries must not invalidate this thread's capture
            mode = dict(capture_error_mode="thread_local") if trainer.comm is not None else {}
            self.graph = torch.cuda.CUDAGraph()
            cap = _capture_stream(self.t.device)
            ops.prepare_streamk_ws(dev, cap)       # the captured stream-K launches' workspace: not from the graph's pool
            if self._pool is not None:
                mode["pool"] = self._pool
            with torch.cuda.graph(self.graph, stream=cap, **mode):
                self.scal = self._body()
        else:
            self.graph = None
            # the step's dropout masks (ops.MaskPlan: one launch per step once the list of sites repeats) are NOT part of
            # the segments: they depend on (seed, site, step counter) only, so the replay draws step k + 1's on the second
            # stream while step k's clip + Adam runs on the first (an integer-multiply-bound ~0.1 ms kernel beside an
            # HBM-bound one) -- into buffers of this object, which the captured layers read
            self._mask_tables, self._mask_event, self.mask_offset = None, None, None
            mp = ops.mask_plan
            if mp.enabled and mp.plan and ops.dropout_state.record is None and \
                    os.environ.get("DV3_MASK_PREDRAW", "1") not in ("0", ""):
                mp.static, self._mask_tables = mp.build(dev, site0)
                self._mask_buffers = mp.static            # (this object keeps the buffers alive: the replays read them)
                self._mask_seed = ops.dropout_state.seed  # the seed the captured single-site launches carry too
            try:
                self._capture_segments()
            finally:
                mp.static = None
            if self._mask_tables is not None:
                torch.cuda.synchronize()
                self.mask_offset = self.seed_offset.clone()      # the counter value the first replay runs under
                mp.draw(self._mask_tables, self.mask_offset, seed=self._mask_seed)
                self._mask_event, self._mask_join = torch.cuda.Event(), torch.cuda.Event()
                self._mask_event.record(torch.cuda.current_stream())
        ops.dropout_state.site = site0

    def _capture_segments(self):
        """capture the step as segments (class docstring).  Backward runs on autograd's worker thread and the segment
        boundaries fall inside it, so every capture is begun in the RELAXED mode (a capture of another mode must be ended
        by the thread that began it)."""
        import gc
        from . import _lib
        t, SS = self.t, ops.SideStream
        side_raw = t.side_stream.cuda_stream
        pool = self._pool if self._pool is not None else torch.cuda.graph_pool_handle()
        st = dict(g=None, side_open=False, forks=0)

        def begin_seg():
            g = torch.cuda.CUDAGraph()
            g.capture_begin(pool=pool, capture_error_mode="relaxed")
            st["g"] = g
            _lib.call("dv3_graph_side_begin", side_raw)
            st["side_open"] = True

        def end_seg():
            ex, n = ctypes.c_void_p(), ctypes.c_int32()
            _lib.call("dv3_graph_side_end", side_raw, ctypes.byref(ex), ctypes.byref(n))
            st["side_open"] = False
            g, st["g"] = st["g"], None
            g.capture_end()
            self.segs.append((g, ex if (ex.value and n.value > 0) else None))
            self.seg_buckets.append(t.comm.take_completed() if t.comm is not None else [])

        flag = self._flag
        timeout_ms = int(os.environ.get("DV3_FLAG_TIMEOUT_MS", "2000"))

        def on_fork_flag():
            # the first fork point ends the graph of zero_grad + forward + losses (the replay orders the side graph after
            # it with an ordinary event, so the side stream does not sit in a wait kernel for the whole forward); from
            # there on both captures run to the end of backward, tied at every fork point by the flag
            st["forks"] += 1
            j = st["forks"]
            if j == 1:
                end_seg()
                begin_seg()
            if j >= 4096:
                raise RuntimeError("GraphedTrainer(flag_sync): more than 4095 fork points in a step")
            p = flag.data_ptr()
            k = self.flag_every
            if k > 1 and self.n_forks >= j:
                # a signal at every k-th fork point only (and at the first and the last): fewer one-thread kernels on the
                # critical queue; a weight gradient then waits for the next signal at or after its own fork point
                if j == 1 or j % k == 0 or j == self.n_forks:
                    _lib.call("dv3_flag_signal", p, p + 8, j, int(j == 1), SS.main.cuda_stream)
                jw = j if j == 1 else min(-(-j // k) * k, self.n_forks)
                _lib.call("dv3_flag_wait", p, p + 16, jw, int(j == 1), p + 24, timeout_ms, side_raw)
                return
            _lib.call("dv3_flag_signal", p, p + 8, j, int(j == 1), SS.main.cuda_stream)
            # the weight gradient of fork point j starts when the step stream has reached fork point j + lag: the
            # input-gradient chain is the critical path, and weight gradients that start the moment their operands exist
            # take compute units from it (the segments' lag of one segment, without their boundaries)
            jw = min(j + self.flag_lag, self.n_forks) if self.n_forks >= j else j
            _lib.call("dv3_flag_wait", p, p + 16, jw, int(j == 1), p + 24, timeout_ms, side_raw)

        def on_fork():
            st["forks"] += 1
            # data parallel (round 6): a bucket that became complete inside this segment ends it -- its all-reduce is
            # issued from the host right after the segment it closes, not `chunk` layers later (nyanko's encoder, 49 MB in
            # five buckets, used to be final only with the last segment: 0.66 ms of exposed wait beside a ring stand-in,
            # profiles/r06_collective_standin.txt)
            bucket_done = t.comm is not None and bool(t.comm._completed) and self.cut_on_bucket
            # the LAST forks one per segment (round 6): side segment j starts when step segment j has run to its end, so the
            # weight-gradient branch of the last segment has nothing left to run beside -- the step stream sat idle for
            # 0.46 ms before clip + Adam while the last four layers' weight gradients ran (profiles/r06a timeline); with
            # one-layer segments at the end only the last layer's is exposed
            tail = self.n_forks > 0 and self.n_forks - st["forks"] < self.tail_fine
            # ... and the FIRST forks one per segment (round 6): segment 0 holds the whole forward, and side segment 0 starts
            # when it ends -- with `chunk` layers of backward inside it the weight gradients of the model's LAST layers (the
            # converter's: the largest of the step) waited for those layers' input-gradient chain to finish, 1.3-1.5 ms in
            # which the step stream ran alone (profiles/r06_split_head.txt)
            head = st["forks"] <= self.head_fine
            if st["forks"] % self.chunk == 0 or bucket_done or tail or head:
                end_seg()
                begin_seg()

        torch.cuda.synchronize()
        gc.collect()
        cap = _capture_stream(self.t.device)
        ops.prepare_streamk_ws(t.device, cap)       # the captured stream-K launches' workspace: not from the graphs' pool
        cap.wait_stream(torch.cuda.current_stream())
        SS.split_capture, SS.split_on_fork = True, (on_fork_flag if self.flag_sync else on_fork)
        try:
            with torch.cuda.stream(cap):
                begin_seg()
                t._zero_grad()
                self.scal = t.forward_backward(self.batch)
                end_seg()
                if t.comm is not None:
                    t.comm.disarm()
                    seen = set(b for bs in self.seg_buckets for b in bs)
                    self.rest_buckets = [b for b in range(len(t.comm.buckets)) if b not in seen]
                g2 = torch.cuda.CUDAGraph()
                g2.capture_begin(pool=pool, capture_error_mode="relaxed")
                try:
                    self._tail(self.scal)
                finally:
                    g2.capture_end()
                self.graph2 = g2
        finally:
            SS.split_capture, SS.split_on_fork = False, None
            if st["side_open"]:          # a failed capture must not leave the side stream capturing
                try:
                    ex, n = ctypes.c_void_p(), ctypes.c_int32()
                    _lib.call("dv3_graph_side_end", side_raw, ctypes.byref(ex), ctypes.byref(n))
                    _lib.call("dv3_graph_destroy", ex)
                except Exception:
                    pass
            if st["g"] is not None:
                try:
                    st["g"].capture_end()
                except Exception:
                    pass
        torch.cuda.current_stream().wait_stream(cap)
        if not any(ex is not None for _, ex in self.segs):
            raise RuntimeError("GraphedTrainer(split_streams): the step has no weight-gradient branch to split off")
        self._seg_events = [torch.cuda.Event() for _ in self.segs]
        self._join_event = torch.cuda.Event()

    def _tail(self, scal):
        t = self.t
        # inside the segmented capture the all-reduces are not part of the tail graph (class docstring)
        t.optimizer_step(reduce=not (self.split and torch.cuda.is_current_stream_capturing()))
        scal["grad_norm"] = t._scalar(t.norm_out[0:1], 1.0 / t.world)
        if ops.gemm_precision() == "f16x3":
            scal["f16_range_events"] = ops.f16_range_events_tensor(t.device)
        self.seed_offset.add_(1)
        return scal

    def _body(self):
        t = self.t
        t._zero_grad()
        scal = t.forward_backward(self.batch)
        return self._tail(scal)

    _TENSORS = ("text", "text_positions", "frame_positions", "mel", "y", "done", "speaker_ids", "input_lengths",
                "target_lengths", "decoder_lengths")

    def load(self, batch):
        """copy `batch` (same shapes as the captured one) into the static batch the graph reads"""
        st = self.batch
        if batch is st:
            return
        for name in self._TENSORS:
            a, b = getattr(st, name), getattr(batch, name)
            if (a is None) != (b is None) or (a is not None and (a.shape != b.shape or a.dtype != b.dtype)):
                raise RuntimeError("GraphedTrainer: batch field %s %s does not fit the captured %s" % (
                    name, None if b is None else tuple(b.shape), None if a is None else tuple(a.shape)))
        va, vb = st.valid, batch.valid
        if (va is None) != (vb is None) or (va is not None and (
                (va.t_in, va.t_dec, va.r, va.downsample_step) != (vb.t_in, vb.t_dec, vb.r, vb.downsample_step) or
                vb.tail_in > va.tail_in or vb.tail_dec > va.tail_dec)):
            raise RuntimeError("GraphedTrainer: the batch's valid lengths do not fit the captured step's")
        self.t.check_lengths(batch)
        for name in self._TENSORS:
            a = getattr(st, name)
            if a is not None:
                a.copy_(getattr(batch, name), non_blocking=True)
        if va is not None:       # the batch's own maxima: the device scalars the captured kernels read
            va.buf.copy_(vb.buf, non_blocking=True)
        st.input_lengths_host, st.target_lengths_host = batch.input_lengths_host, batch.target_lengths_host
        st.decoder_lengths_host, st.n_frames = batch.decoder_lengths_host, batch.n_frames

    def step(self, batch=None):
        if batch is not None:
            self.load(batch)
        self.t._set_hyper()
        if not self.split:
            self.graph.replay()
        else:
            from . import _lib
            cur, side = torch.cuda.current_stream(), self.t.side_stream
            side_raw = side.cuda_stream
            comm = self.t.comm
            if self._mask_tables is not None:
                cur.wait_event(self._mask_event)            # this step's masks (drawn beside the previous step's tail)
            segs = self.segs
            if self.flag_sync:
                # [forward graph] -> event -> the side stream; then [backward graph] on the step stream and [the
                # weight-gradient graph] on the side stream side by side, ordered at every fork point by the device flag
                # (the segments before the last one -- the forward -- go through the loop below as always)
                segs = self.segs[:-1]
            for j, ((g, ex), ev) in enumerate(zip(segs, self._seg_events)):
                g.replay()
                if ex is not None:        # side segment j reads what step segment j wrote: an ordinary event orders them
                    ev.record(cur)
                    side.wait_event(ev)
                    _lib.call("dv3_graph_launch", ex, side_raw)
                if comm is not None and self.seg_buckets[j]:
                    comm.launch_after(self.seg_buckets[j], (cur, side))
            if self.flag_sync:
                g, ex = self.segs[-1]
                ev = self._seg_events[-1]
                ev.record(cur)                 # the end of the forward graph(s)
                side.wait_event(ev)
                g.replay()
                if ex is not None:
                    _lib.call("dv3_graph_launch", ex, side_raw)
            self._join_event.record(side)
            cur.wait_event(self._join_event)
            if comm is not None:
                if self.rest_buckets:
                    comm.launch_after(self.rest_buckets, (cur,))
                comm.join(cur)
            if self._mask_tables is not None:
                # backward has read this step's masks (both streams are joined here): the next step's, on the second stream
                self._mask_join.record(cur)
                side.wait_event(self._mask_join)
                with torch.cuda.stream(side):
                    self.mask_offset.add_(1)
                    ops.mask_plan.draw(self._mask_tables, self.mask_offset, seed=self._mask_seed)
                    self._mask_event.record(side)
            self.graph2.replay()
        ops.bump_param_epoch()           # the replayed clip/Adam wrote the parameters
        self.t.global_step += 1
        if self.flag_sync and not self._flag_checked:
            # once, after the first replay: a wait that gave up means the two streams share a hardware queue (or a signal
            # was lost) -- the step then ran with its weight gradients out of order
            self._flag_checked = True
            n = self.flag_timeouts()
            if n:
                raise RuntimeError("GraphedTrainer(flag_sync): %d fork-point waits timed out in the first replay -- the two "
                                   "backward streams do not run side by side here; set DV3_FLAG_SYNC=0" % n)
        return self.scal

    def flag_timeouts(self):
        """fork-point waits of the replayed steps that gave up (host sync); 0 in a healthy run"""
        if self._flag is None:
            return 0
        return int(self._flag[3].item()) & 0xFFFFFFFF

    def check_range(self):
        """The f16x3 range guard for replayed steps.  A captured graph cannot change its GEMM mode and its clip / Adam
        nodes have already run when the host looks, so the guard here is: call this every N replays (one host sync);
        when operands left the fp16 range on any rank it switches the process to bf16x3 (Trainer.check_range) and
        returns the event count -- the caller must then discard this object (`close()`), restore the parameters from
        its last checkpoint if the step's `grad_norm` was not finite, and capture a new GraphedTrainer (the new
        capture packs bf16x3 operands).  0 = in range, keep replaying."""
        return self.t.check_range()

    def close(self):
        """Give the process-wide dropout state back: the device seed offset installed for the replays would
        otherwise keep shifting the masks of every later (eager) forward in the process."""
        if ops.dropout_state.dev_offset is self.seed_offset:
            ops.dropout_state.dev_offset = self._prev_offset
        self._prev_offset = None
        self._mask_tables = self._mask_buffers = None
        if self.segs:
            from . import _lib
            torch.cuda.synchronize()
            for _, ex in self.segs:
                if ex is not None:
                    _lib.call("dv3_graph_destroy", ex)
            self.segs = []


def clone_batch(batch):
    """a Batch with its own copies of every device tensor (the static batch a captured step reads)"""
    c = lambda t: t.clone() if t is not None else None
    r = batch.mel.shape[1] // batch.frame_positions.shape[1]
    ds = batch.y.shape[1] // batch.mel.shape[1]
    b = Batch(c(batch.text), c(batch.text_positions), c(batch.frame_positions), c(batch.mel), c(batch.y), c(batch.done),
              batch.input_lengths_host.copy(), batch.target_lengths_host.copy(), c(batch.speaker_ids), r, ds,
              batch.text.device)
    if batch.valid is not None:
        b.valid = batch.valid.clone()
    return b


class LatticeReplay(object):
    """Ragged epochs without the host in the loop.  The reference's sampler (train.py:195-239) yields
    about one new padded shape per batch, so a captured step never sees its shape again.  Here a batch is padded to the
    next point of a LATTICE of shapes (data.device_collate(lattice=(step_in, step_dec)): text positions up to a multiple
    of step_in, decoder steps up to a multiple of step_dec) and carries its own maxima as device scalars
    (ops.ValidLengths): the step computes exactly what the reference computes on the batch padded to its own maxima
    (tests/test_gpu_valid_lengths.py), and its shape is one of a few dozen.  One GraphedTrainer per shape, captured the
    first time the shape is met (two forward + backward passes without an update, then the capture), replayed
    afterwards; all of them share one block pool (a step at a time), so the memory is that of the largest shape.
    An LRU bounds the number of live captures.

    step(batch) -> the scalars of the step (device tensors of the captured step that ran; read them before the next
    step of the same shape).  `stats`: captures, replays, capture seconds."""

    def __init__(self, trainer, max_graphs=48, warmup=2):
        import collections
        self.t = trainer
        self.max_graphs, self.warmup = int(max_graphs), int(warmup)
        self.graphs = collections.OrderedDict()
        self.pool = torch.cuda.graph_pool_handle() if trainer.device.type == "cuda" else None
        self.stats = dict(captures=0, replays=0, capture_s=0.0, evictions=0)
        self._n_made = 0

    @staticmethod
    def key_of(batch):
        v = batch.valid
        if v is None:
            raise RuntimeError("LatticeReplay: the batch carries no valid lengths (data.device_collate(lattice=...))")
        return (v.t_in, v.t_dec, int(batch.text.shape[0]))

    def step(self, batch):
        import time
        key = self.key_of(batch)
        g = self.graphs.get(key)
        if g is None:
            t0 = time.perf_counter()
            if len(self.graphs) >= self.max_graphs:
                _, old = self.graphs.popitem(last=False)
                old.close()
                self.stats["evictions"] += 1
            st = ops.dropout_state
            if st.seed is None:
                st.manual_seed(torch.initial_seed())
            seed0 = st.seed
            # every capture draws from its own Philox stream: its step counter starts at zero like every other's
            st.seed = (seed0 + 0x9E3779B1 * (self._n_made + 1)) & 0x7FFFFFFFFFFFFFFF
            try:
                g = GraphedTrainer(self.t, clone_batch(batch), warmup=self.warmup, dry_warmup=True, pool=self.pool)
            finally:
                st.seed = seed0
            self._n_made += 1
            self.graphs[key] = g
            torch.cuda.synchronize()
            self.stats["captures"] += 1
            self.stats["capture_s"] += time.perf_counter() - t0
        else:
            self.graphs.move_to_end(key)
        self.stats["replays"] += 1
        return g.step(batch)

    def check_range(self):
        """the f16x3 range guard (GraphedTrainer.check_range): > 0 = the process has moved to bf16x3, every capture of
        this object packed f16x3 operands and is dropped -- the next step of each shape captures again"""
        n = self.t.check_range()
        if n:
            self.close()
        return n

    def close(self):
        for g in self.graphs.values():
            g.close()
        self.graphs.clear()


# ------------------------------------------------------------------------------------------------
# checkpoint interop with the reference (train.py:788-867): same file name, same dict
#   {"state_dict", "optimizer" (torch.optim.Adam.state_dict() layout), "global_step", "global_epoch"}
# so a run can move between the reference trainer and this one in either direction.
# ------------------------------------------------------------------------------------------------
def checkpoint_dict(trainer, global_epoch=0, save_optimizer_state=True):
    """The dict train.save_checkpoint writes (train.py:800-807), from the flat arena."""
    a, c = trainer.arena, trainer.cfg
    opt = None
    if save_optimizer_state:
        state = {}
        slot = {id(p): (o, n) for o, n, p in zip(a.offsets, a.sizes, a.params)}
        if trainer.adam_step > 0:
            for i, p in enumerate(trainer.optimizer_order):      # torch.optim.Adam's numbering, whatever the arena's order
                if id(p) not in slot:                            # split modes: Adam holds no state for what it never stepped
                    continue
                o, n = slot[id(p)]
                state[i] = dict(step=torch.tensor(float(trainer.adam_step)),
                                exp_avg=a.exp_avg[o:o + n].view(p.shape).clone(),
                                exp_avg_sq=a.exp_avg_sq[o:o + n].view(p.shape).clone())
        group = dict(lr=float(trainer.current_lr()), betas=(c.adam_beta1, c.adam_beta2), eps=c.adam_eps,
                     weight_decay=c.weight_decay, amsgrad=False, maximize=False, foreach=None, capturable=False,
                     differentiable=False, fused=None, params=list(range(len(trainer.optimizer_order))))
        opt = dict(state=state, param_groups=[group])
    return {"state_dict": {k: v.detach().clone() for k, v in trainer.checkpoint_module().state_dict().items()},
            "optimizer": opt, "global_step": trainer.global_step, "global_epoch": global_epoch}


def save_checkpoint(trainer, checkpoint_dir, global_epoch=0, save_optimizer_state=True):
    """train.save_checkpoint (train.py:788-809): checkpoint_step{:09d}.pth in `checkpoint_dir`."""
    import os
    path = os.path.join(checkpoint_dir, "checkpoint_step{:09d}{}.pth".format(trainer.global_step, trainer.checkpoint_suffix()))
    torch.save(checkpoint_dict(trainer, global_epoch, save_optimizer_state), path)
    return path


def _load_file(path, unsafe=False):
    """torch.load restricted to tensors / containers / numbers (what train.save_checkpoint writes); the general
    unpickler (arbitrary code execution from an untrusted file) only behind `unsafe=True`."""
    # a reference run stores numpy scalars (global_step is incremented from numpy values): allow exactly those
    safe = [np.dtype]
    try:
        import numpy._core.multiarray as _ncm        # numpy >= 2
        safe.append(_ncm.scalar)
    except ImportError:
        import numpy.core.multiarray as _ncm         # numpy 1.x
        safe.append(_ncm.scalar)
    safe += [type(np.dtype(t)) for t in (np.int64, np.int32, np.float64, np.float32, np.bool_)]
    safe = [g for g in safe if g is not None]
    try:
        with torch.serialization.safe_globals(safe):
            return torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        if not unsafe:
            raise
        return torch.load(path, map_location="cpu", weights_only=False)


def load_checkpoint(path_or_dict, trainer, reset_optimizer=False, unsafe=False, module=None):
    """train.load_checkpoint (train.py:852-867): model weights, (unless reset_optimizer) the Adam
    moments, and the two counters.  Returns global_epoch.  Accepts files written by the reference:
    its optimizer enumerates get_trainable_parameters() in the same order the arena does.

    module: the sub-module the file's weights belong to -- the reference loads its `_seq2seq` / `_postnet`
    checkpoints into model.seq2seq / model.postnet while training the WHOLE model (train.py:986-990).  None: the
    trainer's own checkpoint module, or, when the file's keys are exactly those of model.seq2seq or model.postnet, that
    sub-module.  The optimizer state of such a partial file covers its own parameters only: the moments of the others
    stay as they are, and since one `adam_step` serves the whole arena the file's step count is taken only when
    EVERY parameter of the arena got a state (otherwise the moments are loaded and the bias corrections keep the
    trainer's own count: pass reset_optimizer=True for the reference's behaviour of a fresh optimizer)."""
    ck = path_or_dict if isinstance(path_or_dict, dict) else _load_file(path_or_dict, unsafe)
    a = trainer.arena
    with torch.no_grad():     # copy INTO the arena views (load_state_dict would keep them too; be explicit)
        target = module if module is not None else trainer.checkpoint_module()
        if module is None and target is trainer.model:
            keys = set(ck["state_dict"])
            for sub in (trainer.model.seq2seq, trainer.model.postnet):
                if keys == set(sub.state_dict()):
                    target = sub
        own = target.state_dict()
        missing = [k for k in own if k not in ck["state_dict"]]
        unexpected = [k for k in ck["state_dict"] if k not in own]
        if missing or unexpected:
            raise RuntimeError("state_dict mismatch: missing %s unexpected %s" % (missing, unexpected))
        for k, v in ck["state_dict"].items():
            own[k].copy_(v)
    opt = ck.get("optimizer")
    if opt is not None and not reset_optimizer:
        if len(opt["param_groups"]) != 1 or len(opt["param_groups"][0]["params"]) != len(trainer.optimizer_order):
            raise RuntimeError("optimizer state does not match get_trainable_parameters()")
        steps = set()
        slot = {id(p): (o, n) for o, n, p in zip(a.offsets, a.sizes, a.params)}
        loaded = 0
        for i, p in enumerate(trainer.optimizer_order):
            st = opt["state"].get(i)
            if st is None or id(p) not in slot:      # (a split-mode trainer keeps the moments of its own part only)
                continue
            o, n = slot[id(p)]
            a.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            a.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
            loaded += 1
        if len(steps) > 1:
            raise RuntimeError("per-parameter Adam step counts differ: %s" % sorted(steps))
        if loaded == len(a.params):
            trainer.adam_step = steps.pop() if steps else 0
        # (a partial state -- a sub-module's checkpoint under a joint trainer: see the docstring)
    trainer.global_step = int(ck["global_step"])
    return int(ck.get("global_epoch", 0))


def _state_dict_of(path_or_dict, unsafe=False):
    ck = path_or_dict if isinstance(path_or_dict, dict) else _load_file(path_or_dict, unsafe)
    return ck["state_dict"] if "state_dict" in ck else ck


def restore_parts(path_or_dict, model):
    """train.restore_parts (train.py:878-897): take from a checkpoint every entry whose name the model
    has and whose shape fits, leave the rest of the model as it is (transfer between presets, e.g. a
    single-speaker seq2seq into a multi-speaker model).  Entries with a different shape are skipped with
    a warning, as the reference does after its per-parameter retry.  Copies INTO the existing tensors,
    so a Trainer's flat parameter arena stays the storage.  -> (restored names, skipped names)"""
    import warnings
    state = _state_dict_of(path_or_dict)
    own = model.state_dict()
    restored, skipped = [], []
    with torch.no_grad():
        for k, v in state.items():
            if k not in own:
                continue
            if tuple(own[k].shape) != tuple(v.shape):
                warnings.warn("%s: may contain invalid size of weight. skipping..." % k)
                skipped.append(k)
                continue
            own[k].copy_(v)
            restored.append(k)
    return restored, skipped


def load_embedding(path_or_dict, model):
    """train._load_embedding (train.py:870-873): the text embedding table of a checkpoint into
    model.seq2seq.encoder.embed_tokens (same KeyError when the checkpoint has none)."""
    state = _state_dict_of(path_or_dict)
    w = state["seq2seq.encoder.embed_tokens.weight"]
    dst = model.seq2seq.encoder.embed_tokens.weight
    if tuple(dst.shape) != tuple(w.shape):
        raise RuntimeError("embedding table %s does not fit %s" % (tuple(w.shape), tuple(dst.shape)))
    with torch.no_grad():
        dst.copy_(w)


def load_submodule_checkpoint(path_or_dict, module):
    """`--checkpoint-seq2seq` / `--checkpoint-postnet` of the reference (train.py:985-989): a checkpoint
    written for model.seq2seq or model.postnet alone goes into that sub-module (strict names, weights
    only -- the reference passes the full model's optimizer there, whose state can not match)."""
    state = _state_dict_of(path_or_dict)
    own = module.state_dict()
    missing = [k for k in own if k not in state]
    unexpected = [k for k in state if k not in own]
    if missing or unexpected:
        raise RuntimeError("state_dict mismatch: missing %s unexpected %s" % (missing, unexpected))
    with torch.no_grad():
        for k, v in state.items():
            own[k].copy_(v)
