# round 2, call 1: the whole GPU suite incl. the preset-scale parity tests, then the bench line as it stands
R=$PWD; mkdir -p gpurun_out; rm -f gpurun_out/parity_scale.jsonl
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_preset_scale.py > gpurun_out/r2a_tests_old.log 2>&1; echo "old tests rc=$?"; tail -3 gpurun_out/r2a_tests_old.log
timeout 1500 python -m pytest tests/test_gpu_preset_scale.py -q > gpurun_out/r2a_tests_scale.log 2>&1; echo "scale tests rc=$?"; tail -40 gpurun_out/r2a_tests_scale.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2a_bench64.log 2>gpurun_out/r2a_bench64.err; echo "b64 rc=$?"; tail -1 gpurun_out/r2a_bench64.log | cut -c1-600
