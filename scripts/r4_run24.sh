#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
bash scripts/r4_prof.sh r04c_vctk --preset deepvoice3_vctk --gemm bf16 --no-graph 2>&1 | tail -42
