# what the driver runs at round end, in one call: smoke, the full GPU suite, the default bench line
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/final_smoke.log
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/final_tests.log
timeout 900 python bench.py > gpurun_out/final_bench.log 2> gpurun_out/final_bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/final_bench.log
