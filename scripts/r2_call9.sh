R=$PWD; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_ddp.py -q -x > gpurun_out/r2j_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2j_tests.log
timeout 900 python -m pytest tests/test_gpu_preset_scale.py -q -x -k "f16x3 or bf16-" > gpurun_out/r2j_scale.log 2>&1; echo "scale rc=$?"; tail -3 gpurun_out/r2j_scale.log
timeout 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline_wgrad']['us_per_launch'], d['roofline_wgrad']['frac'], d['config']['host_enqueue_ms_per_step'])"
