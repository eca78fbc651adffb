R=$PWD; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_ddp.py -q -x > gpurun_out/r2e_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r2e_tests.log
timeout 900 python -m pytest tests/test_gpu_preset_scale.py -q -x -k "f16x3 and (train_step or eval_forward)" > gpurun_out/r2e_scale.log 2>&1; echo "scale rc=$?"; tail -4 gpurun_out/r2e_scale.log
bash scripts/r2_prof.sh r2e 2>&1 | head -24
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-200
