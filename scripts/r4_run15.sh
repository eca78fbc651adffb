#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 300 python scripts/pp2_sk_check.py > gpurun_out/r15_sk.txt 2>&1; echo "rc $?"; cat gpurun_out/r15_sk.txt | tail -20
