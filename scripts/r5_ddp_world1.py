# coding: utf-8
"""Round 5: the three presets' step under a world-size-1 RCCL group (bench.ddp_world1_config) against the same step
without a group, in one process: `collectives issued from the weight-gradient stream (async)` (default) and, with
DV3_COLLECTIVE_STREAM=own, round 4's own collective stream.  Prints the `stream_queues` records."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--presets", default="deepvoice3_ljspeech:f16x3,nyanko_ljspeech:bf16,deepvoice3_vctk:bf16")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
args = argparse.Namespace(batch=64, text_len=150, frames=800, settle=1.0)
bench._world1_group()
out = {}
for item in a.presets.split(","):
    preset, gemm = item.split(":")
    # the no-group step first (replay)
    run = bench.TrainRun(dev, None, 0, 1, preset, gemm, args.batch, args.text_len, args.frames, graph=True)
    m = run.measure(12, 4, settle_s=args.settle)
    run.close()
    r = bench.ddp_world1_config(dev, preset, gemm, args, m["ms_per_step"])
    out[preset + "_" + gemm] = r
    print(preset, gemm, "no group %.3f ms" % m["ms_per_step"],
          {k: (r[k].get("ms_per_step"), r[k].get("vs_no_group")) for k in ("eager", "hipgraph") if k in r}, flush=True)
print(json.dumps(dict(collective_stream=os.environ.get("DV3_COLLECTIVE_STREAM", "issue from the weight-gradient stream"),
                      GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES"), results=out)))
