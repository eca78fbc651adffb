# coding: utf-8
"""Time per decoder step of the free-running decode (slope between two step counts, so the per-utterance setup
drops out) for the persistent program and the launch-by-launch paths.  Developer tool."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deepvoice3_pytorch_amd import builder  # noqa: E402

dev = torch.device("cuda:0")
preset = sys.argv[1] if len(sys.argv) > 1 else "deepvoice3_ljspeech"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
bname, hp, _ = bench.PRESETS[preset]
torch.manual_seed(0)
model = getattr(builder, bname)(**hp).to(dev).eval()
model.make_generation_fast_()
dec = model.seq2seq.decoder
Tt = 100
rng = np.random.RandomState(0)
text = torch.from_numpy(rng.randint(2, hp["n_vocab"], (B, Tt))).to(dev)
tpos = torch.arange(1, Tt + 1).repeat(B, 1).to(dev)
spk = torch.zeros(B, dtype=torch.long, device=dev) if hp.get("n_speakers", 1) > 1 else None
with torch.no_grad():
    se = model.embed_speakers(spk) if spk is not None else None
    enc = model.seq2seq.encoder(text, lengths=None, speaker_embed=se)
kw = dict(speaker_embed=se) if bname == "deepvoice3" else {}


def run(n):
    dec.min_decoder_steps = dec.max_decoder_steps = n
    with torch.no_grad():
        for _ in range(2):
            dec.incremental_forward(enc, tpos, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            dec.incremental_forward(enc, tpos, **kw)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3


for label, persistent, graph, abl in (("library-launched chunks", "launched", False, "0"), ("persistent", True, False, "0"), ("persistent, acquire by every wave", True, False, "16"),
                                      ("persistent, no release fence", True, False, "8"),
                                      ("persistent, no fences", True, False, "1"),
                                      ("persistent, no barriers", True, False, "7"),
                                      ("launches, step graph", False, True, "0"), ("launches, eager", False, False, "0")):
    os.environ["DV3_DECODE_ABLATE"] = abl
    dec.persistent_decode, dec.use_step_graph = persistent is True, graph
    dec.launched_decode = persistent == "launched"
    a, b = run(40), run(200)
    print("%-36s %s B=%d: %.1f us per decoder step (setup %.2f ms)" % (label, preset, B, (b - a) / 160 * 1e6,
                                                                      (a - 41 * (b - a) / 160) * 1e3), flush=True)
