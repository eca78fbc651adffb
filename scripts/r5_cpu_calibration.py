# coding: utf-8
"""Build container only (round 5): the reference's own train.train() step (train.py:604-785, imported
unmodified through oracle/refimport.py) against the oracle port, at the headline batch (64) and at the preset's own
batch (16), on all host cores and on one thread, with the share of the pure-Python guided_attention
(train.py:586-601: what an install without numba pays) timed separately.
Writes profiles/r05_cpu_port_vs_reference.json; bench.py's cpu_baseline quotes its B=64 all-core ratio.

    python scripts/r5_cpu_calibration.py            # ~15-25 min on an 8-core container
"""
import json
import os
import platform
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def guided_attention_seconds(B, Tt, n_frames):
    """the reference's guided_attentions() for one batch of the workload (decoder lengths = frames / 4)"""
    from oracle import refimport
    train, hparams, _ = refimport.load_train_module()
    il = np.full((B,), Tt, dtype=np.int64)
    tl = np.full((B,), n_frames // 4, dtype=np.int64)
    t0 = time.time()
    train.guided_attentions(il, tl, int(tl.max()), g=0.2)
    return time.time() - t0


def one(B, threads, max_seconds):
    torch.set_num_threads(threads)
    ref = bench.cpu_baseline_reference(B, 150, 800, max_seconds=max_seconds)
    ga = guided_attention_seconds(B, 150, 800)
    port = bench._PortStep(B, 150, 800)
    dt, n = port.time(3, max_seconds)
    step_ref = B * 800.0 / ref["value"]
    return dict(batch=B, threads=threads, reference_frames_per_s=ref["value"], reference_s_per_step=round(step_ref, 3),
                reference_sample=ref["sample"], guided_attention_s_per_step=round(ga, 3),
                guided_attention_share_of_reference_step=round(ga / step_ref, 4),
                port_frames_per_s=round(port.frames / dt, 1), port_s_per_step=round(dt, 3), port_steps=n,
                port_over_reference_time=round(dt / step_ref, 4))


if __name__ == "__main__":
    ncpu = os.cpu_count() or 1
    runs = []
    for B in (64, 16):
        for threads in (ncpu, 1):
            r = one(B, threads, max_seconds=90.0 if threads > 1 else 200.0)
            print(json.dumps(r), flush=True)
            runs.append(r)
    head = [r for r in runs if r["batch"] == 64 and r["threads"] == ncpu][0]
    out = dict(host_cpus=ncpu, cpu_model=cpu_model(), torch=torch.__version__, runs=runs,
               port_over_reference_time=head["port_over_reference_time"],
               note="step time of the oracle port / step time of the reference's own train.train() on the same host and "
                    "workload (B items of Tt=150 / 800 frames); the reference runs without numba, i.e. its "
                    "guided_attention is a Python double loop (guided_attention_share_of_reference_step)")
    json.dump(out, open(os.path.join(bench.ROOT, "profiles", "r05_cpu_port_vs_reference.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
