#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ddp.py -q -x -m gpu 2>&1 | tail -5
for i in 1 2 3; do
timeout 500 python scripts/r4_group_replay_check.py > gpurun_out/r13_group_$i.txt 2>&1; echo "rc $?"; grep "eager\|replay\|rror" gpurun_out/r13_group_$i.txt | cut -c1-330
done
