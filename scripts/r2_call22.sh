mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -k "bf16 and not bf16x3" > gpurun_out/r2s_tests_bf16.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r2s_tests_bf16.log
timeout 900 python -m pytest tests/test_gpu_c8.py -m gpu -q > gpurun_out/r2s_tests_c8.log 2>&1; echo "c8 tests rc=$?"; tail -8 gpurun_out/r2s_tests_c8.log
