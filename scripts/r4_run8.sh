#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/r4_prof.sh r04a --no-graph 2>&1 | tail -42 > gpurun_out/r04a_prof_summary.txt
BENCH_ARGS="" bash scripts/r4_prof.sh r04b --no-graph --preset nyanko_ljspeech --gemm bf16 2>&1 | tail -42 > gpurun_out/r04b_prof_summary.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/pmc_r4.sh 2>&1 | tail -120 > gpurun_out/r04_pmc_summary.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/r04a_prof gpurun_out/r04b_prof gpurun_out/pmc_r4_*/ 2>/dev/null
head -30 gpurun_out/r04a_prof_summary.txt; head -30 gpurun_out/r04b_prof_summary.txt; tail -60 gpurun_out/r04_pmc_summary.txt
