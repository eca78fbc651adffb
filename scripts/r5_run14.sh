#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for P in eager graph nosk graph+graph eager+graph; do timeout 300 python scripts/r5_nan_hunt3.py $P 2>&1 | grep -v amdgpu.ids | tail -5; done
