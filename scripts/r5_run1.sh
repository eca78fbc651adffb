#!/bin/bash
# round 5, call 1: LOAD-phase structure variants of the 256 x 256 tap-GEMM (bit-identity + timing)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
DV3_LIBPATH=$PWD/deepvoice3_pytorch_amd/libdv3hip_exp.so timeout 900 python scripts/r5_pp2_ord.py > gpurun_out/r5_pp2_ord.txt 2>&1; echo "rc $?"
tail -60 gpurun_out/r5_pp2_ord.txt
