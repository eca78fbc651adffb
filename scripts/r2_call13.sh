timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "fused_attention" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_model.py -q -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_preset_scale.py -q -x -k "f16x3 and (train_step or eval_forward)" 2>&1 | tail -2
for f in 1 0; do DV3_FUSED_ATTN=$f timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('FUSED_ATTN=$f', d['value'], d['ms_per_step'])"; done
