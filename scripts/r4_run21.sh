#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 400 python scripts/small_gemm_tiles.py 2>&1 | tail -9 | tee gpurun_out/r21_small.txt
