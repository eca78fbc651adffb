#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "stream_k or pingpong" 2>&1 | tail -5
timeout 900 python -m pytest tests -q -x -m gpu > gpurun_out/r18_tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r18_tests.log
timeout 300 python scripts/r4_sk_step_ab.py 2>&1 | tail -3
