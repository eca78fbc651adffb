#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python scripts/r4_spk_step_ab.py 2>&1 | tail -5 | tee gpurun_out/r26_spk_ab.txt
