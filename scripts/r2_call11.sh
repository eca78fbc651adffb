timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_preset_scale.py -q -x -k "bf16" 2>&1 | tail -3
for p in nyanko_ljspeech deepvoice3_vctk; do
timeout 300 python bench.py --preset $p --gemm bf16 --no-extras --no-cpu-baseline --no-roofline --steps 30 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$p bf16', d['value'], d['ms_per_step'])"
done
