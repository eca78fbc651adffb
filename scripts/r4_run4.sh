#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ddp.py tests/test_gpu_range_guard.py tests/test_gpu_c8.py -q -m gpu -k "train_step or graphed_train or resume or save_load or priority or ddp or world1 or ranks or trainer_reports or tracks_fp32 or batched_weight" 2>&1 | tail -12 ) > $O/r4_4_tests.log
( timeout 200 python scripts/aten_in_step.py deepvoice3_ljspeech:f16x3 2>&1 | grep -v amdgpu.ids ) > $O/r4_4_aten.txt
( timeout -s KILL 240 python scripts/two_graph_probe.py 2>&1 | grep -v amdgpu.ids ) > $O/r4_4_two_graph.txt
cat $O/r4_4_tests.log; cut -c1-200 $O/r4_4_aten.txt; cat $O/r4_4_two_graph.txt
