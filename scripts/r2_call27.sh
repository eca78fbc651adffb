mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_c8.py -m gpu -q > gpurun_out/r2x_tests_c8.log 2>&1; echo "c8 tests rc=$?"; tail -8 gpurun_out/r2x_tests_c8.log
timeout 900 python -m pytest tests -m gpu -q -k "bf16 and not bf16x3" > gpurun_out/r2x_tests_bf16.log 2>&1; echo "bf16 tests rc=$?"; tail -4 gpurun_out/r2x_tests_bf16.log
for pr in nyanko_ljspeech deepvoice3_vctk; do
  timeout 300 python bench.py --preset $pr --gemm bf16 --no-extras --no-cpu-baseline --no-roofline --steps 20 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c8 $pr', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"
done
