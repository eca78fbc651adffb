#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python scripts/r5_prio_ab.py > gpurun_out/r5_prio_ab.txt 2>&1; echo "rc $?"; grep -v "amdgpu.ids" gpurun_out/r5_prio_ab.txt | tail -8
