mkdir -p gpurun_out
timeout 700 python scripts/c8_check.py > gpurun_out/r2s_c8.log 2>&1; echo "c8 rc=$?"; grep -v "^$" gpurun_out/r2s_c8.log | grep -v "^Conv1dGLU\|^HighwayConv1d\|roundtrip\|mask_bits\|padding" | tail -40
timeout 900 python -m pytest tests -m gpu -q -k "bf16 and not bf16x3" > gpurun_out/r2s_tests_bf16.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r2s_tests_bf16.log
for st in 1 0; do for pr in nyanko_ljspeech deepvoice3_vctk; do
  DV3_BF16_STORAGE=$st timeout 300 python bench.py --preset $pr --gemm bf16 --no-extras --no-cpu-baseline --no-roofline --steps 20 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('storage=$st $pr', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"
done; done
