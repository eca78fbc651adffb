# c8 storage: unit tests, the bf16 preset-scale tests and the two bf16 bench lines
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_c8.py -m gpu -q 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -q -k "bf16 and not bf16x3" 2>&1 | tail -2
for pr in nyanko_ljspeech deepvoice3_vctk; do
  timeout 300 python bench.py --preset $pr --gemm bf16 --no-extras --no-cpu-baseline --no-roofline --steps 20 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c8 $pr', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"
done
