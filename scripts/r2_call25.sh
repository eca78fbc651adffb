mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_audio.py tests/test_gpu_c8.py -m gpu -q > gpurun_out/r2v_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r2v_tests.log
timeout 300 python bench.py --mode synth --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-1200
timeout 300 python bench.py --preset nyanko_ljspeech --gemm bf16 --no-extras --no-cpu-baseline --no-roofline --steps 20 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c8 nyanko', d['value'], d['ms_per_step'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
