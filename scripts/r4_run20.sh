#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 400 python scripts/r4_sk_step_ab.py 2>&1 | tail -4
