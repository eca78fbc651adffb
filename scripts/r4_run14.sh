#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 300 python scripts/layer_times.py 64 > gpurun_out/r14_layers.txt 2>&1; cat gpurun_out/r14_layers.txt | tail -60
