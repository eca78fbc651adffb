#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_speaker_bias.py -q -x 2>&1 | tail -6
timeout 300 python scripts/spk_bias_time.py 2>&1 | tail -3
