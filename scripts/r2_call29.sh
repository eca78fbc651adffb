mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_c8.py -m gpu -q -k "converters" 2>&1 | tail -2
for pr in nyanko_ljspeech deepvoice3_vctk; do
  timeout 300 python bench.py --preset $pr --gemm bf16 --no-extras --no-cpu-baseline --no-roofline --steps 20 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c8 $pr', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"
done
BENCH_ARGS="--preset nyanko_ljspeech --gemm bf16" bash scripts/r2_prof.sh r2t_nyanko_bf16 | head -16
