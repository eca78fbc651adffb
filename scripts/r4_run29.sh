#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 300 python scripts/spk_bias_time.py 2>&1 | tail -4
