#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_gate_fuse.py -q 2>&1 | tail -30 > gpurun_out/r6_run1_gatefuse.txt
cat gpurun_out/r6_run1_gatefuse.txt
timeout 1200 python scripts/r6_gate_fuse_step_ab.py 2>&1 | grep -v Warning | tail -8 > gpurun_out/r6_gate_fuse_step_ab.txt
cat gpurun_out/r6_gate_fuse_step_ab.txt
