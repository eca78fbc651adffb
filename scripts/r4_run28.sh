#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_speaker_bias.py -q -x 2>&1 | tail -3
bash scripts/r4_prof.sh r04c_vctk --preset deepvoice3_vctk --gemm bf16 --no-graph 2>&1 | grep -i "spk\|total kernel"
