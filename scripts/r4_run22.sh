#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 400 python scripts/small_gemm_tiles.py 2>&1 | tail -8 | cut -c1-70
timeout 400 python scripts/r4_j1_step_ab.py 2>&1 | tail -2
