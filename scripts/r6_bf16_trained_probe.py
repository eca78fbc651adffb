# coding: utf-8
"""Round 6: where does the bf16 eval forward of the TRAINED preset-width deepvoice3_ljspeech leave the f16x3 one?
(tests/test_gpu_bf16_trained.py measured linear 0.166 / alignments 0.11 of the tensor maximum, mel 7.5e-3.)  Trains as the
test does, then compares the forwards output by output with max / rms / quantiles, isolates the converter (both modes on
the SAME decoder states) and separates operand rounding from storage rounding (bf16 GEMMs on fp32 storage)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from tests import test_gpu_bf16_trained as T
from deepvoice3_pytorch_amd import ops, train_step, builder

dev = torch.device("cuda:0")
bname, hp, ga = bench.PRESETS["deepvoice3_ljspeech"]
hp = dict(hp)
batches = T._batches(hp)
torch.manual_seed(7)
sd0 = {k: v.detach().clone() for k, v in builder.deepvoice3(**hp).state_dict().items()}
model, tr = T._trainer("f16x3", hp, ga, sd0, None, dev)
dbs = T._dev_batches(train_step, batches, dev)
ops.dropout_state.manual_seed(41)
for i in range(T.WARM):
    tr.step(dbs[i % T.N_BATCH])
sd1 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
tr.close()
bt = batches[0]
mel_in = bt["mel"][:, 0::4, :].contiguous().to(dev)


def fwd(mode, storage, states=None):
    ops.set_gemm_precision(mode)
    ops.bf16_storage = storage
    m = builder.deepvoice3(**hp)
    m.load_state_dict(sd1)
    m.to(dev).eval()
    with torch.no_grad():
        if states is not None:
            return m.postnet(states).float().cpu().numpy().astype(np.float64)
        enc = m.seq2seq.encoder(bt["text"].to(dev), lengths=bt["input_lengths"])
        mel, ali, done, st = m.seq2seq.decoder(enc, mel_in, text_positions=bt["text_positions"].to(dev),
                                               frame_positions=bt["frame_positions"].to(dev), lengths=bt["input_lengths"])
        lin = m.postnet(st)
    return dict(keys=enc[0], values=enc[1], mel=mel, ali=ali, done=done, states=st, linear=lin)


def stats(name, a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    e = np.abs(a - b)
    sc = max(np.abs(b).max(), 1e-30)
    q = np.quantile(e, [0.5, 0.9, 0.99, 0.999, 0.9999]) / sc
    print("  %-8s max %.3g  rms %.3g  of max|ref| %.3g | quantiles 50/90/99/99.9/99.99%%: %s | share of elements off by > 1e-2: %.2e"
          % (name, e.max() / sc, np.sqrt((e ** 2).mean()) / sc, sc, " ".join("%.2e" % x for x in q), float((e / sc > 1e-2).mean())))


ref = fwd("f16x3", True)
for tag, mode, storage in (("bf16 GEMMs, bf16 c8 storage (the benchmarked form)", "bf16", True),
                           ("bf16 GEMMs, fp32 storage", "bf16", False), ("bf16x3 (bf16 pairs everywhere)", "bf16x3", True)):
    got = fwd(mode, storage)
    print(tag)
    for k in ("keys", "values", "states", "mel", "ali", "done", "linear"):
        stats(k, got[k].float().cpu().numpy(), ref[k].float().cpu().numpy())
    lin_same_states = fwd(mode, storage, states=ref["states"])
    stats("converter alone (on the f16x3 decoder states)", lin_same_states, ref["linear"].float().cpu().numpy())
ops.set_gemm_precision("f16x3")
# how big are the converter's pre-sigmoid logits at these weights?  (an output error e needs a logit error >= 4 e)
y = ref["linear"].float().clamp(1e-6, 1 - 1e-6)
logit = torch.log(y / (1 - y))
print("converter logits at the trained weights: |logit| mean %.2f max %.2f" % (float(logit.abs().mean()), float(logit.abs().max())))
