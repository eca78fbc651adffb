#!/bin/bash
# round 4, final sequence: full GPU test tier, smoke, the bench line, kernel stats of the headline step, per-layer table
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r32_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r32_tests.log
tail -4 gpurun_out/r32_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r32_smoke.log 2>&1; tail -2 gpurun_out/r32_smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r32_bench.json 2> gpurun_out/r32_bench.err; echo "bench rc $?"
tail -c 600 gpurun_out/r32_bench.json
bash scripts/r4_prof.sh r04c_train --no-graph 2>&1 | head -16
timeout 200 python scripts/layer_times.py 64 > gpurun_out/r32_layers.txt 2>&1; head -12 gpurun_out/r32_layers.txt
