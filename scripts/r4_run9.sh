#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout -s KILL 500 python scripts/split_chunk_ab.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r4_9_chunk.txt
cat gpurun_out/r4_9_chunk.txt
