R=$PWD; mkdir -p gpurun_out; rm -f gpurun_out/parity_scale.jsonl
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_preset_scale.py > gpurun_out/r2i_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r2i_tests.log
timeout 1500 python -m pytest tests/test_gpu_preset_scale.py -q > gpurun_out/r2i_scale.log 2>&1; echo "scale rc=$?"; tail -6 gpurun_out/r2i_scale.log
timeout 900 python bench.py > gpurun_out/r2i_bench.log 2>gpurun_out/r2i_bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r2i_bench.err; tail -1 gpurun_out/r2i_bench.log | cut -c1-1500
