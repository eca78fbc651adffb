# coding: utf-8
"""Round 6: what one GPU can measure about the 8-GPU step's gradient all-reduce.

The data-parallel step's bucket schedule (dist.BucketedAllReduce: buckets cut from the arena's tail, issued from the
weight-gradient stream as soon as their last gradient is final, awaited before clip + Adam) with dist.RingStandin in the
communicator's place: `channels` persistent workgroups of 256 threads that stream 2 (n-1)/n of every bucket through
HBM at the pace an n-rank xGMI ring would set.  Sweeps: channels (RCCL's NCCL_MAX_NCHANNELS), the assumed all-reduce
bus bandwidth, the bucket size; three presets at per-GPU batch 64 and 16.  Prints step time without a group, beside
the stand-in, the exposed wait, and the weak-scaling efficiency the pair predicts.

Also prints what RCCL itself reports for a world-size-1 group (NCCL_DEBUG=INFO: channel count lines), for the record.
argv: quick  -> the default point only"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse
import torch
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
args = argparse.Namespace(text_len=150, frames=800, settle=0.5, batch=64)

CASES = (("deepvoice3_ljspeech", "f16x3", 64), ("deepvoice3_ljspeech", "f16x3", 16), ("nyanko_ljspeech", "bf16", 64),
         ("deepvoice3_vctk", "bf16", 64), ("nyanko_ljspeech", "bf16", 16), ("deepvoice3_vctk", "bf16", 16))
for preset, gemm, B in CASES:
    run = bench.TrainRun(dev, None, 0, 1, preset, gemm, B, 150, 800, graph=True)
    base = run.measure(12, 4, settle_s=0.5)["ms_per_step"]
    run.close()
    print("%s %s B=%d: no group %.3f ms/step" % (preset, gemm, B, base), flush=True)
    points = [dict()]
    if not quick:
        points += [dict(channels=c) for c in (4, 8, 32, 64)] + [dict(busbw_gbps=b) for b in (75.0, 300.0)] + \
                  [dict(bucket_mb=m) for m in (8.0, 50.0)] + [dict(threads=512)]
    for kw in points:
        r = bench.ddp_standin_config(dev, preset, gemm, B, args, base, **kw)
        if "error" in r:
            print("   ", kw, r)
            continue
        print("    channels %2d x %d thr, busbw %5.0f GB/s, buckets %s MB: %.3f ms/step (x%.4f), exposed %.3f ms, wire %.3f ms "
              "-> predicted 8-GPU weak-scaling efficiency %.3f | issued after segments %s"
              % (r["channels"], r["threads"], r["busbw_gbps"], r["bucket_mb"], r["ms_per_step"], r["step_inflation"],
                 r["allreduce_exposed_ms"] or 0.0, r["wire_ms_per_step"], r["predicted_weak_scaling_efficiency"],
                 r["buckets_issued_after_segment"]), flush=True)

# what RCCL reports for its own kernels on this box (a world-size-1 group has a degenerate topology: for the record only)
code = ("import os,torch,torch.distributed as d;os.environ['HSA_ENABLE_IPC_MODE_LEGACY']='0';"
        "d.init_process_group('nccl',init_method='tcp://127.0.0.1:29571',rank=0,world_size=1);"
        "t=torch.ones(1<<22,device='cuda');d.all_reduce(t);torch.cuda.synchronize();d.destroy_process_group()")
env = dict(os.environ, NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH,TUNING")
try:
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    lines = [l for l in (out.stdout + out.stderr).splitlines() if any(k in l for k in ("hannel", "threads", "RCCL version", "nranks", "Algo"))]
    print("RCCL (world 1, NCCL_DEBUG=INFO):")
    for l in lines[:25]:
        print("   ", l[:200])
except Exception as e:
    print("RCCL probe failed:", e)
