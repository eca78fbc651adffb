# end-of-round check after the ping-pong main loop: tests, the bench line, the in-phase loop beside it, kernel stats
R=$PWD; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r1e_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r1e_tests.log
python bench.py > gpurun_out/r1e_bench64.log 2>gpurun_out/r1e_bench64.err; echo "b64 rc=$?"; tail -c 300 gpurun_out/r1e_bench64.err
DV3_X3_PINGPONG=0 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r1e_bench64_inphase.log 2>&1; echo "b64 in-phase rc=$?"
python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r1e_bench64_b.log 2>&1; echo "b64 (2nd) rc=$?"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1e_prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r1e_prof.log 2>&1; echo "prof rc=$?"
cd $R; for f in gpurun_out/r1e_bench64.log gpurun_out/r1e_bench64_inphase.log gpurun_out/r1e_bench64_b.log; do tail -1 $f | cut -c1-400; done
