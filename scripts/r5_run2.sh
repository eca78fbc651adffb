#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
ls /sys/class/drm/ 2>/dev/null | head; 
DV3_LIBPATH=$PWD/deepvoice3_pytorch_amd/libdv3hip_exp.so timeout 600 python scripts/r5_pp2_power.py > gpurun_out/r5_pp2_power.txt 2>&1; echo "rc $?"
tail -40 gpurun_out/r5_pp2_power.txt
