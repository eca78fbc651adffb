#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for P in normal low; do DV3_SIDE_PRIORITY=$P timeout 600 python scripts/r5_nan_hunt.py 2>&1 | grep -v amdgpu.ids | tail -7; done
