#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
cd scripts && DV3_LIBPATH=$PWD/../deepvoice3_pytorch_amd/libdv3hip_exp.so timeout 900 python r5_pp2_cvc.py > ../gpurun_out/r5_pp2_cvc.txt 2>&1; echo "rc $?"
tail -60 ../gpurun_out/r5_pp2_cvc.txt
