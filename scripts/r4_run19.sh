#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "stream_k" 2>&1 | tail -3
timeout 300 python scripts/layer_times.py 64 > gpurun_out/r19_layers.txt 2>&1; grep "150 3\|total" gpurun_out/r19_layers.txt | head -20
