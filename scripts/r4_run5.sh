#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 0 1 2; do ( timeout -s KILL 150 python scripts/two_graph_probe.py $i 2>&1 | grep -v "amdgpu.ids\|^  \|^    \|^Search\|^HIP kernel\|^For debug\|^Compile" ) >> gpurun_out/r4_5_two_graph.txt; done
cat gpurun_out/r4_5_two_graph.txt
