#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python scripts/r4_vctk_group.py both > gpurun_out/r11_vctk.txt 2>&1; cat gpurun_out/r11_vctk.txt | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
