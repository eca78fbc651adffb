#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 400 python scripts/tile_sweep_j3.py 2>&1 | tail -13 | tee gpurun_out/r23_j3.txt
