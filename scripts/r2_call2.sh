# round 2, call 2: f16x3 default: whole GPU suite, preset-scale parity, new bench line
R=$PWD; mkdir -p gpurun_out; rm -f gpurun_out/parity_scale.jsonl
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_preset_scale.py > gpurun_out/r2b_tests_old.log 2>&1; echo "old tests rc=$?"; tail -15 gpurun_out/r2b_tests_old.log
timeout 1500 python -m pytest tests/test_gpu_preset_scale.py -q > gpurun_out/r2b_tests_scale.log 2>&1; echo "scale tests rc=$?"; tail -25 gpurun_out/r2b_tests_scale.log
timeout 900 python bench.py > gpurun_out/r2b_bench64.log 2>gpurun_out/r2b_bench64.err; echo "b64 rc=$?"; tail -c 1500 gpurun_out/r2b_bench64.err; tail -1 gpurun_out/r2b_bench64.log | cut -c1-3000
