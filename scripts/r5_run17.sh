#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
cd scripts && timeout 900 python r5_c8pp_rl.py > ../gpurun_out/r5_c8pp_rl.txt 2>&1; echo "rc $?"; grep -v amdgpu.ids ../gpurun_out/r5_c8pp_rl.txt | tail -24
