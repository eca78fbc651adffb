# coding: utf-8
"""Build container only: time the reference's own train.train() (bench.cpu_baseline_reference) and the oracle
port (bench.cpu_baseline_port) on the same host, same workload shape (B items of Tt=150 / 800 frames), and record
the ratio in profiles/r02_cpu_port_vs_reference.json -- on the GPU box /root/reference does not exist, so bench.py
times the port there and quotes this ratio beside it."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ref = bench.cpu_baseline_reference(B, 150, 800, max_seconds=60.0)
port = bench.cpu_baseline_port(B, 150, 800, max_seconds=60.0)
out = dict(batch=B, host_cpus=os.cpu_count(), threads=torch.get_num_threads(), reference=ref, port=port,
           port_over_reference_time=round(ref["value"] / port["value"], 4),
           note="ratio of step times port/reference = reference rate / port rate; > 1 means the port is slower")
json.dump(out, open(os.path.join(bench.ROOT, "profiles", "r02_cpu_port_vs_reference.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
