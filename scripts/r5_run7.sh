#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
cd scripts && timeout 600 python r5_c8pp_rf_debug.py 2>&1 | tail -40
