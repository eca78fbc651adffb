mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/t6_all.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/t6_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/t6_smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/t6_smoke.log
timeout 900 python bench.py > gpurun_out/t6_bench.log 2>&1; echo "bench rc $?"; tail -1 gpurun_out/t6_bench.log | cut -c1-600
