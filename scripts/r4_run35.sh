#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_model.py -q -x -k "ddp or world or refetch or split_stream or graphed or golden or rank" 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r35_bench.json 2> gpurun_out/r35_bench.err; echo "bench rc $?"
tail -c 200 gpurun_out/r35_bench.json
