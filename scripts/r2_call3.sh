# round 2, call 3: planes kernel: A/B + bit-equality at the north-star shape, then the test suites and the step with DV3_PLANES=1
R=$PWD; mkdir -p gpurun_out
timeout 600 python scripts/planes_ab.py > gpurun_out/r2c_planes_ab.log 2>&1; echo "ab rc=$?"; tail -20 gpurun_out/r2c_planes_ab.log
DV3_PLANES=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x > gpurun_out/r2c_tests_planes.log 2>&1; echo "planes tests rc=$?"; tail -8 gpurun_out/r2c_tests_planes.log
DV3_PLANES=1 timeout 600 python bench.py --no-extras --no-cpu-baseline --no-roofline > gpurun_out/r2c_bench_planes.log 2>&1; echo "bench planes rc=$?"; tail -1 gpurun_out/r2c_bench_planes.log | cut -c1-400
