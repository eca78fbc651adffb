#!/bin/bash
# round 4, closing check of the final tree: full GPU test tier, smoke, bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r34_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r34_tests.log
tail -4 gpurun_out/r34_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r34_smoke.log 2>&1; tail -2 gpurun_out/r34_smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r34_bench.json 2> gpurun_out/r34_bench.err; echo "bench rc $?"
tail -c 300 gpurun_out/r34_bench.json
