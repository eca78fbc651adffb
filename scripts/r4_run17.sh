#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 300 python scripts/pp2_sk_check.py > gpurun_out/r17_sk.txt 2>&1; echo "rc $?"; tail -10 gpurun_out/r17_sk.txt
timeout 300 python scripts/pp2_sk_abl.py > gpurun_out/r17_skabl.txt 2>&1; echo "rc $?"; tail -6 gpurun_out/r17_skabl.txt
