# coding: utf-8
"""Per-GEMM-shape time table of one eager train step (B from argv, default 64): which layer shapes the
tap-GEMM / wgrad kernels spend the step on.  Measurement aid, not a test."""
import sys, os, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from deepvoice3_pytorch_amd import builder, train_step, ops, _lib
_L = _lib.lib()

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
hp = dict(bench.DV3_LJ)
torch.manual_seed(0)
model = builder.deepvoice3(**hp).to(dev)
trainer = train_step.Trainer(model, train_step.TrainConfig(max_positions=hp["max_positions"]))
bt = bench.synth_batch(np.random.RandomState(1), B, 150, 800, hp)
batch = train_step.Batch.from_collate(bt["text"], bt["input_lengths"], bt["mel"], bt["y"], bt["text_positions"],
                                      bt["frame_positions"], bt["done"], bt["target_lengths"], None,
                                      downsample_step=4, device=dev)
for _ in range(2):
    trainer.step(batch)
torch.cuda.synchronize()
rec = collections.defaultdict(lambda: [0, 0.0, 0.0])
orig_conv, orig_wgrad = ops.conv_gemm, ops.wgrad_gemm


def timed(fn, key, flops, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn(*a, **k)
    e1.record()
    torch.cuda.synchronize()
    key = key + ("v%d" % _L.dv3_debug_get(10),)
    r = rec[key]
    r[0] += 1; r[1] += e0.elapsed_time(e1) * 1e3; r[2] += flops
    return out


def conv(x, a, lda, a_half, **k):
    key = ("conv", "dgrad" if k.get("mode") == ops.EPI_DGRAD else "fwd", k["B"], k["Cin"], k["M"], k["Tout"], k.get("J", 1),
           k.get("dil", 1), "x3" if k.get("a_split") is not None else "f32", "mask" if (k.get("xmask") is not None or k.get("ymask") is not None) else "")
    return timed(orig_conv, key, 2.0 * k["B"] * k["Tout"] * k["M"] * k["Cin"] * k.get("J", 1), x, a, lda, a_half, **k)


def wgrad(g, x, **k):
    key = ("wgrad", "", k["B"], k["Cin"], k["M"], k["T"], k.get("J", 1), k.get("dil", 1), "x3" if k.get("split_bf16") else "f32",
           "S%d" % k.get("n_slabs", 1))
    return timed(orig_wgrad, key, 2.0 * k["B"] * k["T"] * k["M"] * k["Cin"] * k.get("J", 1), g, x, **k)


ops.conv_gemm, ops.wgrad_gemm = conv, wgrad
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
trainer.step(batch)
torch.cuda.synchronize()
tot = sum(r[1] for r in rec.values())
print("GEMM launches %d, total %.2f ms" % (sum(r[0] for r in rec.values()), tot / 1e3))
print("%-70s %5s %9s %8s %7s" % ("kind dir B Cin M T J dil arith flags", "n", "total us", "avg us", "TF/s"))
for key, r in sorted(rec.items(), key=lambda kv: -kv[1][1]):
    print("%-70s %5d %9.0f %8.1f %7.1f" % (" ".join(str(v) for v in key), r[0], r[1], r[1] / r[0], r[2] / r[1] / 1e6))
