#!/bin/bash
# round 5: full GPU tier (tests, smoke, default bench line) on the current tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu -x > gpurun_out/r5_tests_c.log 2>&1; echo "tests rc $?" >> gpurun_out/r5_tests_c.log
tail -5 gpurun_out/r5_tests_c.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5_smoke_c.log 2>&1; tail -2 gpurun_out/r5_smoke_c.log
timeout 900 python bench.py > gpurun_out/r5_bench_c.json 2> gpurun_out/r5_bench_c.err; echo "bench rc $?"; tail -c 600 gpurun_out/r5_bench_c.json
