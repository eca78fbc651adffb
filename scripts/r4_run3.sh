#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 300 python scripts/aten_in_step.py 2>&1 | grep -v amdgpu.ids ) > $O/r4_3_aten.txt
( time timeout 400 python -m pytest tests/test_audio.py tests/test_gpu_training_curve.py tests/test_gpu_model.py -q -m gpu -k "spectrogram or trajector or train_step_matches_reference_golden" 2>&1 | tail -8 ) > $O/r4_3_tests.log 2>&1
cat $O/r4_3_aten.txt | cut -c1-260; cat $O/r4_3_tests.log
