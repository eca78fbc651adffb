mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r2w_bench.log 2> gpurun_out/r2w_bench.err; echo "bench rc=$?"; tail -c 2500 gpurun_out/r2w_bench.log
bash scripts/r2_prof_synth.sh | head -14
timeout 300 python bench.py --preset nyanko_ljspeech --gemm bf16 --graph --no-extras --no-cpu-baseline --no-roofline --steps 20 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c8 nyanko graph', d['value'], d['ms_per_step'])"
