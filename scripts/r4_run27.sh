#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_speaker_bias.py -q -x 2>&1 | tail -4
timeout 600 python scripts/r4_spk_step_ab.py 2>&1 | tail -4 | tee gpurun_out/r27_spk_ab.txt
bash scripts/r4_prof.sh r04c_vctk --preset deepvoice3_vctk --gemm bf16 --no-graph 2>&1 | grep -i "spk\|total kernel"
