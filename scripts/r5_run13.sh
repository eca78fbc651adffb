#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python scripts/r5_nan_hunt2.py 2>&1 | grep -v amdgpu.ids | tail -34
