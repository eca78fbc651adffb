mkdir -p gpurun_out
timeout 800 python scripts/offset_repro.py > gpurun_out/r2r_offset.log 2>&1; echo "rc=$?"; grep "^==" gpurun_out/r2r_offset.log; tail -3 gpurun_out/r2r_offset.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_preset_scale.py -m gpu -x -q -k "graphed or train_step_matches_oracle" > gpurun_out/r2r_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r2r_tests.log
