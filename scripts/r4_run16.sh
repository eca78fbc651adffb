#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 300 python scripts/pp2_sk_abl.py > gpurun_out/r16_skabl.txt 2>&1; echo "rc $?"; tail -8 gpurun_out/r16_skabl.txt
