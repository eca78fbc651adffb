#!/bin/bash
# round 5: stream-K (reversed indices, per-XCD groups, workspace outside captures), c8pp pin at the north star, bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_c8.py -q -m gpu -x -k "stream_k or north_star or c8pp" > gpurun_out/r5_tests_d.log 2>&1; echo "tests rc $?" >> gpurun_out/r5_tests_d.log
tail -6 gpurun_out/r5_tests_d.log
timeout 300 python scripts/pp2_sk_check.py > gpurun_out/r5_pp2_sk_check.txt 2>&1; tail -25 gpurun_out/r5_pp2_sk_check.txt
timeout 900 python bench.py > gpurun_out/r5_bench_d.json 2> gpurun_out/r5_bench_d.err; echo "bench rc $?"; tail -c 300 gpurun_out/r5_bench_d.json
