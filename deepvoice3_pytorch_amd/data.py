# coding: utf-8
"""Batch assembly in the conventions of the reference's train.collate_fn (train.py:293-360), so a
training loop can feed train_step.Trainer from the reference's own datasets: ragged (text ids, mel,
linear[, speaker id]) items -> padded tensors + position tensors + done flags.

`collate_fn` is host-side (numpy) like the reference; the masks and guided-attention weights that the
reference builds on the host afterwards (train.py:261-271,594-601) are computed on the device by the
loss kernels from the length vectors this function returns.

`pack_batch` + `device_collate` are the device-side form of the same assembly (SURVEY section 8f, rank
1): the host only concatenates the ragged items (no padding loops, no padded bytes over PCIe: three
flat buffers + two offset vectors instead of nine padded tensors) and the HIP path pads, down-samples
the mel and derives the position tensors and done flags from the lengths.  Bit-identical to
`to_device_batch(collate_fn(batch))` (tests/test_gpu_model.py).
"""
import numpy as np
import torch

from .train_step import Batch


def collate_fn(batch, outputs_per_step=1, downsample_step=4):
    """-> (x, input_lengths, mel, y, (text_positions, frame_positions), done, target_lengths,
    speaker_ids), exactly the reference's tuple (train.py:359-360)."""
    r, ds = int(outputs_per_step), int(downsample_step)
    n = len(batch)
    multi_speaker = len(batch[0]) == 4
    in_len = np.array([len(item[0]) for item in batch], dtype=np.int64)
    tgt_len = np.array([len(item[1]) for item in batch], dtype=np.int64)
    # frames: round the longest target up to r and to downsample_step, then b_pad = r leading
    # zero frames per decoder stride ("imitates initial decoder states", train.py:313-316)
    T = int(tgt_len.max())
    T += (-T) % r
    T += (-T) % ds
    b_pad = r
    T += b_pad * ds
    Tt = int(in_len.max())
    x = np.zeros((n, Tt), dtype=np.int64)
    tpos = np.zeros((n, Tt), dtype=np.int64)
    mel = np.zeros((n, T, batch[0][1].shape[1]), dtype=np.float32)
    y = np.zeros((n, T, batch[0][2].shape[1]), dtype=np.float32)
    Td = T // r // ds
    done = np.ones((n, Td, 1), dtype=np.float32)
    for i, item in enumerate(batch):
        L, F = int(in_len[i]), int(tgt_len[i])
        x[i, :L] = item[0]
        tpos[i, :L] = np.arange(1, L + 1)
        mel[i, b_pad:b_pad + F] = item[1]
        y[i, b_pad:b_pad + F] = item[2]
        done[i, :max(F // r // ds - 1, 0)] = 0.0
    fpos = np.tile(np.arange(1, Td + 1, dtype=np.int64)[None], (n, 1))
    spk = torch.from_numpy(np.array([item[3] for item in batch], dtype=np.int64)) if multi_speaker else None
    return (torch.from_numpy(x), torch.from_numpy(in_len), torch.from_numpy(mel), torch.from_numpy(y),
            (torch.from_numpy(tpos), torch.from_numpy(fpos)), torch.from_numpy(done), torch.from_numpy(tgt_len), spk)


def to_device_batch(collated, device, outputs_per_step=1, downsample_step=4):
    """collate_fn's tuple -> train_step.Batch on `device` (mel time-downsampled as train.py:639-640)."""
    x, in_len, mel, y, (tpos, fpos), done, tgt_len, spk = collated
    return Batch.from_collate(x, in_len, mel, y, tpos, fpos, done, tgt_len, spk, downsample_step, device,
                              r=outputs_per_step)


class PackedBatch(object):
    """Ragged items of one batch, concatenated on the host (pinned when CUDA is present)."""

    def __init__(self, text, mel, lin, in_len, tgt_len, speaker_ids):
        self.text, self.mel, self.lin = text, mel, lin
        self.in_len, self.tgt_len, self.speaker_ids = in_len, tgt_len, speaker_ids


def pack_batch(batch, pin=None):
    """[(text ids, mel (n, num_mels), linear (n, fft/2+1)[, speaker id])] -> PackedBatch: the items
    back to back in three flat buffers, no padding."""
    in_len = np.array([len(item[0]) for item in batch], dtype=np.int64)
    tgt_len = np.array([len(item[1]) for item in batch], dtype=np.int64)
    for item in batch:
        if len(item[2]) != len(item[1]):
            raise ValueError("mel and linear spectrogram of an item differ in frame count")
    text = torch.from_numpy(np.concatenate([np.asarray(item[0], dtype=np.int64) for item in batch]))
    mel = torch.from_numpy(np.ascontiguousarray(np.concatenate([item[1] for item in batch], 0), dtype=np.float32))
    lin = torch.from_numpy(np.ascontiguousarray(np.concatenate([item[2] for item in batch], 0), dtype=np.float32))
    if pin is None:
        pin = torch.cuda.is_available()
    if pin:
        text, mel, lin = text.pin_memory(), mel.pin_memory(), lin.pin_memory()
    spk = np.array([item[3] for item in batch], dtype=np.int64) if len(batch[0]) == 4 else None
    return PackedBatch(text, mel, lin, in_len, tgt_len, spk)


def padded_frames(tgt_len, outputs_per_step=1, downsample_step=4):
    """(T, b_pad): frame count of the padded batch and its leading zero frames (train.py:307-316)."""
    r, ds = int(outputs_per_step), int(downsample_step)
    T = int(np.max(tgt_len))
    T += (-T) % r
    T += (-T) % ds
    b_pad = r
    return T + b_pad * ds, b_pad


def lattice_shape(max_in, max_dec, step_in=16, step_dec=8):
    """(longest text, most decoder steps) of a batch rounded up to the lattice a LatticeReplay keeps captured steps for
    -> (t_in, t_dec).  The surplus is at most step - 1 columns per axis."""
    up = lambda v, s: -(-int(v) // int(s)) * int(s)
    return up(max_in, step_in), up(max_dec, step_dec)


def device_collate(packed, device, outputs_per_step=1, downsample_step=4, lattice=None):
    """PackedBatch -> train_step.Batch on `device`, equal bit for bit to
    `to_device_batch(collate_fn(batch), device, ...)`; the padding runs on the GPU
    (dv3_ragged_pad_rows_b32), positions and done flags come from the length vectors.
    lattice = (step_in, step_dec): pad to lattice_shape(...) instead of the batch's own maxima and attach those maxima
    as `batch.valid` (ops.ValidLengths) -- a step on such a batch computes what the step on the batch padded to its own
    maxima computes (train_step.Trainer reads `valid`), and its shape is one of a small set (train_step.LatticeReplay)."""
    from . import ops
    r, ds = int(outputs_per_step), int(downsample_step)
    n = len(packed.in_len)
    T, b_pad = padded_frames(packed.tgt_len, r, ds)
    Tt = int(packed.in_len.max())
    Td = T // r // ds
    dev = torch.device(device)
    if lattice is not None:
        return _device_collate_lattice(packed, dev, r, ds, T, b_pad, Tt, Td, lattice)
    f = lambda t: t.to(dev, non_blocking=True)
    text, mel, lin = f(packed.text), f(packed.mel), f(packed.lin)
    off = lambda lens: f(torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)))
    toff, foff = off(packed.in_len), off(packed.tgt_len)
    x = ops.ragged_pad_rows(text, toff, n, Tt, lead=0, t_stride=1)
    y = ops.ragged_pad_rows(lin, foff, n, T, lead=b_pad, t_stride=1)
    # the reference down-samples the padded mel in time (train.py:639-640): same rows, picked while padding
    mel_ds = ops.ragged_pad_rows(mel, foff, n, (T + ds - 1) // ds if ds > 1 else T, lead=b_pad, t_stride=ds)
    in_len_d = f(torch.from_numpy(packed.in_len))
    ar = torch.arange(1, Tt + 1, device=dev, dtype=torch.int64)[None]
    tpos = ar * (ar <= in_len_d[:, None])
    fpos = torch.arange(1, Td + 1, device=dev, dtype=torch.int64)[None].repeat(n, 1)
    first_done = f(torch.from_numpy(np.maximum(packed.tgt_len // r // ds - 1, 0)))
    done = (torch.arange(Td, device=dev)[None] >= first_done[:, None]).float()[:, :, None]
    spk = f(torch.from_numpy(packed.speaker_ids)) if packed.speaker_ids is not None else None
    return Batch(x, tpos, fpos, mel_ds, y, done, packed.in_len, packed.tgt_len, spk, r, ds, dev)


def _device_collate_lattice(packed, dev, r, ds, T, b_pad, Tt, Td, lattice):
    from . import ops
    n = len(packed.in_len)
    step_in, step_dec = lattice
    t_in, t_dec = lattice_shape(Tt, Td, step_in, step_dec)
    TL = t_dec * r * ds
    f = lambda t: t.to(dev, non_blocking=True)
    text, mel, lin = f(packed.text), f(packed.mel), f(packed.lin)
    off = lambda lens: f(torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)))
    toff, foff = off(packed.in_len), off(packed.tgt_len)
    x = ops.ragged_pad_rows(text, toff, n, t_in, lead=0, t_stride=1)
    y = ops.ragged_pad_rows(lin, foff, n, TL, lead=b_pad, t_stride=1)
    mel_ds = ops.ragged_pad_rows(mel, foff, n, TL // ds, lead=b_pad, t_stride=ds)
    in_len_d = f(torch.from_numpy(packed.in_len))
    ar = torch.arange(1, t_in + 1, device=dev, dtype=torch.int64)[None]
    tpos = ar * (ar <= in_len_d[:, None])
    fr = torch.arange(1, t_dec + 1, device=dev, dtype=torch.int64)
    fpos = (fr * (fr <= Td))[None].repeat(n, 1)          # surplus steps: position 0 (no frame of the batch reads them)
    first_done = f(torch.from_numpy(np.maximum(packed.tgt_len // r // ds - 1, 0)))
    done = (torch.arange(t_dec, device=dev)[None] >= first_done[:, None]).float()[:, :, None]
    spk = f(torch.from_numpy(packed.speaker_ids)) if packed.speaker_ids is not None else None
    b = Batch(x, tpos, fpos, mel_ds, y, done, packed.in_len, packed.tgt_len, spk, r, ds, dev)
    b.valid = ops.ValidLengths.make(Tt, Td, n, t_in, t_dec, step_in - 1, step_dec - 1, r, ds, dev)
    return b


def pad_to_shape(batch, t_in, t_dec, tail_in=None, tail_dec=None):
    """A Batch padded to its own maxima -> the same batch padded to (t_in text positions, t_dec decoder steps) with
    `valid` attached (see device_collate(lattice=)).  tail_*: the promised upper bound of the surplus per axis (default:
    the surplus itself)."""
    from . import ops
    B, Tt = batch.text.shape
    Td = batch.frame_positions.shape[1]
    r = batch.mel.shape[1] // Td
    ds = batch.y.shape[1] // batch.mel.shape[1]
    if t_in < Tt or t_dec < Td:
        raise ValueError("pad_to_shape: (%d, %d) is smaller than the batch (%d, %d)" % (t_in, t_dec, Tt, Td))

    def pad(t, n, dim=1):
        if t is None or t.shape[dim] == n:
            return t
        shape = list(t.shape)
        shape[dim] = n
        out = t.new_zeros(shape)
        out.narrow(dim, 0, t.shape[dim]).copy_(t)
        return out

    done = pad(batch.done, t_dec)
    if t_dec > Td:
        done[:, Td:] = 1.0
    b = Batch(pad(batch.text, t_in), pad(batch.text_positions, t_in), pad(batch.frame_positions, t_dec),
              pad(batch.mel, t_dec * r), pad(batch.y, t_dec * r * ds), done, batch.input_lengths_host,
              batch.target_lengths_host, batch.speaker_ids, r, ds, batch.text.device)
    b.valid = ops.ValidLengths.make(Tt, Td, B, t_in, t_dec, t_in - Tt if tail_in is None else tail_in,
                                    t_dec - Td if tail_dec is None else tail_dec, r, ds, batch.text.device)
    return b


class PreprocessedDataset(torch.utils.data.Dataset):
    """A directory written by the reference's preprocess.py (preprocess.py:27-31): `train.txt` with one
    `spec.npy|mel.npy|n_frames|text[|speaker_id]` line per utterance.  Items are the tuples collate_fn
    takes: (text ids, mel (n_frames, num_mels), linear (n_frames, fft/2+1)[, speaker id]) -- what the
    reference assembles from TextDataSource / MelSpecDataSource / LinearSpecDataSource
    (train.py:96-192) through nnmnkwii.  `text_to_sequence` is the text frontend (the reference's
    `frontend.en.text_to_sequence`, out of scope here); `speaker_id` filters a multi-speaker set down
    to one speaker exactly as the reference does (train.py:113-119,163-171)."""

    def __init__(self, data_root, text_to_sequence, speaker_id=None):
        import os
        self.data_root = data_root
        self.text_to_sequence = text_to_sequence
        with open(os.path.join(data_root, "train.txt"), "rb") as f:
            rows = [ln.decode("utf-8").rstrip("\n").split("|") for ln in f if ln.strip()]
        if not rows or len(rows[0]) not in (4, 5):
            raise ValueError("train.txt: expected 4 or 5 '|'-separated columns")
        self.multi_speaker = len(rows[0]) == 5
        if self.multi_speaker and speaker_id is not None:
            rows = [r for r in rows if int(r[-1]) == speaker_id]
            self.multi_speaker = False
        self.rows = rows
        self.frame_lengths = [int(r[2]) for r in rows]      # what the length-bucketing sampler sorts by

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        import os
        r = self.rows[i]
        text = np.asarray(self.text_to_sequence(r[3]), dtype=np.int32)
        mel = np.load(os.path.join(self.data_root, r[1]))
        spec = np.load(os.path.join(self.data_root, r[0]))
        if self.multi_speaker:
            return text, mel, spec, int(r[4])
        return text, mel, spec


class ListDataset(torch.utils.data.Dataset):
    """Items held in host memory (synthetic benchmarks, tests); `repeat` makes the list appear that many
    times longer without copying."""

    def __init__(self, items, repeat=1):
        self.items, self.repeat = list(items), int(repeat)
        self.frame_lengths = [len(it[1]) for it in self.items] * self.repeat

    def __len__(self):
        return len(self.items) * self.repeat

    def __getitem__(self, i):
        return self.items[i % len(self.items)]


class LengthBucketedSampler(object):
    """The reference's PartialyRandomizedSimilarTimeLengthSampler (train.py:195-239) as a BATCH sampler
    that is rank-aware: (1) sort by length, (2) shuffle inside groups of `batch_group_size`, (3) permute
    whole mini-batches, (4) shuffle the tail that does not fill a group -- then cut the index stream into
    consecutive mini-batches exactly as the reference's DataLoader(batch_size=...) does, and give rank r the
    batches r, r + world, r + 2*world, ... (the reference sampler is not rank-aware).  Every
    rank draws the same permutation (numpy RandomState(seed + epoch)), so the shards are disjoint and cover
    the epoch; the batch list is truncated to a multiple of `world` so all ranks take the same number of
    steps (a data-parallel step is collective).  Under world > 1 a short tail batch would meet full batches in
    one 1/world gradient average and over-weight its samples, so drop_last defaults to True there (None = that
    rule); the batches cut off by the truncation rotate with the epoch, so no sample is skipped every epoch."""

    def __init__(self, lengths, batch_size=16, batch_group_size=None, permutate=True, rank=0, world=1, seed=0,
                 drop_last=None):
        lengths = np.asarray(lengths, dtype=np.int64)
        self.sorted_indices = np.argsort(lengths, kind="stable")
        self.batch_size = int(batch_size)
        n = len(lengths)
        if batch_group_size is None:
            batch_group_size = min(self.batch_size * 32, n)
            if batch_group_size % self.batch_size != 0:
                batch_group_size -= batch_group_size % self.batch_size
        if batch_group_size <= 0 or batch_group_size % self.batch_size != 0:
            raise ValueError("batch_group_size must be a positive multiple of batch_size")
        self.batch_group_size = batch_group_size
        self.permutate = permutate
        self.rank, self.world, self.seed, self.epoch = int(rank), int(world), int(seed), 0
        self.drop_last = (int(world) > 1) if drop_last is None else bool(drop_last)
        if not 0 <= self.rank < self.world:
            raise ValueError("rank must be in [0, world)")

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def epoch_batches(self):
        """all mini-batches of the epoch (every rank computes the same list)"""
        rs = np.random.RandomState(self.seed + self.epoch)
        idx = self.sorted_indices.copy()
        g, bs = self.batch_group_size, self.batch_size
        e = 0
        for i in range(len(idx) // g):
            s, e = i * g, (i + 1) * g
            rs.shuffle(idx[s:e])
        if self.permutate and e > 0:
            perm = rs.permutation(e // bs)
            idx[:e] = idx[:e].reshape(-1, bs)[perm].reshape(-1)
        if e < len(idx):
            rs.shuffle(idx[e:])
        batches = [idx[i:i + bs] for i in range(0, len(idx), bs)]
        if self.drop_last and batches and len(batches[-1]) < bs:
            batches.pop()
        return batches

    def __iter__(self):
        batches = self.epoch_batches()
        extra = len(batches) % self.world
        if extra:      # drop `extra` batches at an epoch-dependent position (same on every rank; epoch 0: the last ones)
            nb = len(batches)
            gone = set((nb - extra * (self.epoch + 1) + k) % nb for k in range(extra))
            batches = [b for i, b in enumerate(batches) if i not in gone]
        for b in batches[self.rank::self.world]:
            yield [int(i) for i in b]

    def __len__(self):
        n = len(self.sorted_indices)
        nb = n // self.batch_size if self.drop_last else -(-n // self.batch_size)
        return nb // self.world


class _Staging(object):
    """one pinned staging slot: three flat host buffers that grow on demand"""

    def __init__(self, pin):
        self.pin, self.text, self.mel, self.lin, self.event = pin, None, None, None, None

    def _fit(self, cur, shape, dtype):
        n = int(np.prod(shape))
        if cur is None or cur.numel() < n:
            cur = torch.empty(int(n * 1.25) + 16, dtype=dtype)
            if self.pin:
                cur = cur.pin_memory()
        return cur

    def fill(self, items, pool):
        in_len = np.array([len(it[0]) for it in items], dtype=np.int64)
        tgt_len = np.array([len(it[1]) for it in items], dtype=np.int64)
        for it in items:
            if len(it[2]) != len(it[1]):
                raise ValueError("mel and linear spectrogram of an item differ in frame count")
        nm, nl = items[0][1].shape[1], items[0][2].shape[1]
        nt, nf = int(in_len.sum()), int(tgt_len.sum())
        self.text = self._fit(self.text, (nt,), torch.int64)
        self.mel = self._fit(self.mel, (nf, nm), torch.float32)
        self.lin = self._fit(self.lin, (nf, nl), torch.float32)
        tv, mv, lv = self.text.numpy(), self.mel.numpy()[:nf * nm].reshape(nf, nm), self.lin.numpy()[:nf * nl].reshape(nf, nl)
        to, fo = np.concatenate([[0], np.cumsum(in_len)]), np.concatenate([[0], np.cumsum(tgt_len)])

        def put(i):
            it = items[i]
            tv[to[i]:to[i + 1]] = it[0]
            mv[fo[i]:fo[i + 1]] = it[1]
            lv[fo[i]:fo[i + 1]] = it[2]          # the big one: numpy releases the GIL for the copy
        if pool is not None:
            list(pool.map(put, range(len(items))))
        else:
            for i in range(len(items)):
                put(i)
        spk = np.array([it[3] for it in items], dtype=np.int64) if len(items[0]) == 4 else None
        return PackedBatch(self.text[:nt], self.mel[:nf * nm].view(nf, nm), self.lin[:nf * nl].view(nf, nl),
                           in_len, tgt_len, spk)


class Prefetcher(object):
    """Feeds train_step.Trainer.step with device-resident batches while the previous step computes
    (the reference: DataLoader workers + 8 blocking .to(device) copies per step, train.py:619-663).
    A producer thread walks the sampler; worker threads read the items and copy them back to back into a
    pinned staging slot (no padded bytes cross PCIe); the H2D copies and the device-side collate
    (dv3_ragged_pad_rows, positions, done flags) run on a side HIP stream; `depth` batches are kept in
    flight.  next() makes the consumer's stream wait on the batch's event -- no host synchronisation.
    lattice = (step_in, step_dec): batches padded to a lattice of shapes with their maxima attached
    (device_collate(lattice=)), what train_step.LatticeReplay.step takes."""

    def __init__(self, dataset, batch_sampler, device, outputs_per_step=1, downsample_step=4, depth=2, workers=2,
                 loop=False, beside=None, lattice=None):
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor
        self.dataset, self.sampler = dataset, batch_sampler
        self.device = torch.device(device)
        self.r, self.ds = int(outputs_per_step), int(downsample_step)
        self.loop = loop
        self.lattice = tuple(lattice) if lattice is not None else None
        self.cuda = self.device.type == "cuda"
        # the copy / collate stream: one that shares no hardware queue with the consumer's streams (`beside`: default the
        # current stream; pass the trainer's side stream too) -- see ops.concurrent_stream
        self.side = None
        if self.cuda:
            from . import ops as _ops
            with torch.cuda.device(self.device):
                self.side = _ops.concurrent_stream(list(beside) if beside else [torch.cuda.current_stream()], role="prefetch")
        self.q = queue.Queue(maxsize=max(1, depth))
        self.slots = [_Staging(self.cuda) for _ in range(max(1, depth) + 2)]
        self.pool = ThreadPoolExecutor(max(1, workers)) if workers > 0 else None
        self._stop = threading.Event()
        self._err = None
        self.thread = threading.Thread(target=self._produce, name="dv3-prefetch", daemon=True)
        self.thread.start()

    def _produce(self):
        try:
            if self.cuda:
                torch.cuda.set_device(self.device)
            epoch, k = 0, 0
            while not self._stop.is_set():
                if hasattr(self.sampler, "set_epoch"):
                    self.sampler.set_epoch(epoch)
                for idx in self.sampler:
                    if self._stop.is_set():
                        return
                    slot = self.slots[k % len(self.slots)]
                    k += 1
                    if slot.event is not None:
                        slot.event.synchronize()         # the copies that read this slot have run
                    if self.pool is not None:
                        items = list(self.pool.map(self.dataset.__getitem__, idx))
                    else:
                        items = [self.dataset[i] for i in idx]
                    packed = slot.fill(items, self.pool)
                    if self.cuda:
                        with torch.cuda.stream(self.side):
                            batch = device_collate(packed, self.device, self.r, self.ds, lattice=self.lattice)
                            ev = torch.cuda.Event()
                            ev.record(self.side)
                        slot.event = ev
                    else:
                        batch, ev = device_collate(packed, self.device, self.r, self.ds, lattice=self.lattice), None
                    while not self._stop.is_set():
                        try:
                            self.q.put((batch, ev), timeout=0.1)
                            break
                        except Exception:
                            continue
                epoch += 1
                if not self.loop:
                    break
            self.q.put(None)
        except BaseException as e:      # surface in the consumer
            self._err = e
            try:
                self.q.put_nowait(None)
            except Exception:
                pass

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if item is None:
            if self._err is not None:
                raise self._err
            raise StopIteration
        batch, ev = item
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in (batch.text, batch.text_positions, batch.frame_positions, batch.mel, batch.y, batch.done,
                      batch.input_lengths, batch.target_lengths, batch.decoder_lengths, batch.speaker_ids):
                if t is not None:
                    t.record_stream(cur)         # allocated on the side stream, consumed here
            if batch.valid is not None:
                batch.valid.buf.record_stream(cur)
        return batch

    def close(self):
        self._stop.set()
        try:
            while True:
                self.q.get_nowait()
        except Exception:
            pass
        self.thread.join(timeout=5)
        if self.pool is not None:
            self.pool.shutdown(wait=False)
