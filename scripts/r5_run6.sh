#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
cd scripts && timeout 1200 python r5_ship_check.py > ../gpurun_out/r5_ship_check.txt 2>&1; echo "rc $?"
tail -75 ../gpurun_out/r5_ship_check.txt
