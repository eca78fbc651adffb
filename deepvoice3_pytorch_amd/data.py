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


def device_collate(packed, device, outputs_per_step=1, downsample_step=4):
    """PackedBatch -> train_step.Batch on `device`, equal bit for bit to
    `to_device_batch(collate_fn(batch), device, ...)`; the padding runs on the GPU
    (dv3_ragged_pad_rows_b32), positions and done flags come from the length vectors."""
    from . import ops
    r, ds = int(outputs_per_step), int(downsample_step)
    n = len(packed.in_len)
    T, b_pad = padded_frames(packed.tgt_len, r, ds)
    Tt = int(packed.in_len.max())
    Td = T // r // ds
    dev = torch.device(device)
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
