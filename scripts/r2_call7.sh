R=$PWD; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -q -x -k "decode or incremental or torch_ops or resume or save_load or generation_fast" > gpurun_out/r2g_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r2g_tests.log
timeout 300 python bench.py --mode synth --steps 20 --warmup 3 2>/dev/null | cut -c1-900
