#!/bin/bash
# round 4, GPU call 2: A/B of the kernel switches on whole steps, c8pp ablations, the tests changed since call 1
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 500 python -m pytest tests/test_audio.py tests/test_gpu_c8.py tests/test_gpu_training_curve.py -q -m gpu -k "lws or audio or spectrogram or inv_spec or wgrad_c8 or trajector or griffin" 2>&1 | tail -25 ) > $O/r4_2_tests.log
( timeout 300 python -m pytest tests/test_gpu_model.py -q -m gpu -k "train_step_matches_reference_golden" 2>&1 | tail -25 ) >> $O/r4_2_tests.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) >> $O/r4_2_tests.log
( DV3_LIBPATH=libdv3hip_exp.so timeout 200 python scripts/c8pp_abl.py 2>&1 | grep -v amdgpu.ids ) > $O/r4_2_c8pp_abl.txt
( timeout 400 python scripts/r4_ab_step.py 2>&1 | grep -v amdgpu.ids ) > $O/r4_2_ab_step.txt
cat $O/r4_2_tests.log | tail -40; cat $O/r4_2_c8pp_abl.txt; cat $O/r4_2_ab_step.txt
