#!/bin/bash
# round 4, GPU call 1: everything changed so far, measured once
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 600 python -m pytest tests/test_gpu_c8.py -q -m gpu -k "c8pp" -x 2>&1 | tail -25 ) > $O/r4_1_c8pp_tests.log
( timeout 300 python scripts/c8pp_sweep.py 2>&1 | grep -v amdgpu.ids ) > $O/r4_1_c8pp_sweep.txt
( timeout 200 python scripts/r4_slab_rows.py 2>&1 | grep -v amdgpu.ids ) > $O/r4_1_slab_rows.txt
( DV3_LIBPATH=libdv3hip_exp.so timeout 200 python scripts/r4_abl11.py 2>&1 | grep -v amdgpu.ids ) > $O/r4_1_abl11.txt
( timeout 900 python -m pytest tests -q -m gpu --maxfail=15 2>&1 | tail -40 ) > $O/r4_1_all_tests.log
( timeout 600 python bench.py --steps 20 --warmup 5 > $O/r4_1_bench.json ) 2> $O/r4_1_bench.err
tail -3 $O/r4_1_c8pp_tests.log; cat $O/r4_1_c8pp_sweep.txt; cat $O/r4_1_slab_rows.txt; cat $O/r4_1_abl11.txt; tail -12 $O/r4_1_all_tests.log; tail -c 600 $O/r4_1_bench.err
