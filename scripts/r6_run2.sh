#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > gpurun_out/r6_smoke.txt; cat gpurun_out/r6_smoke.txt
python -m pytest tests/test_gpu_bf16_trained.py tests/test_gpu_c8.py -q -k "trained or same_rounding_oracle" 2>&1 | tail -30 > gpurun_out/r6_run2_tests.txt
cat gpurun_out/r6_run2_tests.txt
cat gpurun_out/bf16_trained.json
timeout 1500 python scripts/r6_collective_standin.py quick 2>&1 | grep -v Warning | tail -40 > gpurun_out/r6_collective_standin_quick.txt
cat gpurun_out/r6_collective_standin_quick.txt
