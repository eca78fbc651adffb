# round-2 state check: smoke, full GPU suite, full bench line, kernel-trace stats of the step and of synthesis
R=$PWD; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2m_smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2m_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2m_tests.log
timeout 600 python bench.py > gpurun_out/r2m_bench.log 2> gpurun_out/r2m_bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/r2m_bench.log
bash scripts/r2_prof.sh r2m
bash scripts/r2_prof_synth.sh
