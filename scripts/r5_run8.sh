#!/bin/bash
# round 5: GPU test tier + default bench line with the new LOAD-phase defaults
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r5_tests_a.log 2>&1; echo "tests rc $?" >> gpurun_out/r5_tests_a.log
tail -5 gpurun_out/r5_tests_a.log
timeout 900 python bench.py > gpurun_out/r5_bench_a.json 2> gpurun_out/r5_bench_a.err; echo "bench rc $?"; tail -c 1500 gpurun_out/r5_bench_a.json
