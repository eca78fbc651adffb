# coding: utf-8
"""Phase stamps of the persistent decode program (workgroup 0, step 8): where an entry's time goes.  Developer tool."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deepvoice3_pytorch_amd import builder, _lib  # noqa: E402

dev = torch.device("cuda:0")
preset, B = "deepvoice3_ljspeech", int(sys.argv[1]) if len(sys.argv) > 1 else 64
bname, hp, _ = bench.PRESETS[preset]
torch.manual_seed(0)
model = getattr(builder, bname)(**hp).to(dev).eval()
model.make_generation_fast_()
dec = model.seq2seq.decoder
rng = np.random.RandomState(0)
text = torch.from_numpy(rng.randint(2, hp["n_vocab"], (B, 100))).to(dev)
tpos = torch.arange(1, 101).repeat(B, 1).to(dev)
with torch.no_grad():
    enc = model.seq2seq.encoder(text, lengths=None, speaker_embed=None)
dec.min_decoder_steps = dec.max_decoder_steps = 30
dec.persistent_decode = True
for abl in (64, 64 + 1, 64 + 7):
    os.environ["DV3_DECODE_ABLATE"] = str(abl)
    with torch.no_grad():
        for _ in range(2):
            dec.incremental_forward(enc, tpos)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (40 * 8))()
    _lib.call("dv3_debug_read", 3, buf, ctypes.sizeof(buf))
    st = np.array(buf, dtype=np.int64).reshape(40, 8)
    print("ablate=%d  (cycles of s_memtime; columns: prefetch+stage issue->landed | sync | gemv | reduce | tail | "
          "next-prefetch issue | barrier)" % abl)
    for e in range(17):
        r = st[e]
        if r[0] == 0:
            print("  entry %2d: (attention or no tile)  barrier %6d" % (e, r[7] - r[6]))
            continue
        print("  entry %2d: %6d %6d %6d %6d %6d %6d %6d   total %6d" % (
            e, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5], r[7] - r[6], r[7] - r[0]))
