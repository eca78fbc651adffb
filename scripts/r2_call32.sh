mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r2z_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2z_tests.log
timeout 900 python bench.py > gpurun_out/r2z_bench.log 2> gpurun_out/r2z_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r2z_bench.log
