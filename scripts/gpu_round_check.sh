R=$PWD
python -m pytest tests -m gpu -x -q > gpurun_out/r1_tests.log 2>&1; echo "tests rc=$?"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r1_bench16.log 2>gpurun_out/r1_bench16.err; echo "b16 rc=$?"; tail -c 600 gpurun_out/r1_bench16.err
python bench.py --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r1_bench64.log 2>&1; echo "b64 rc=$?"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1_prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r1_prof.log 2>&1; echo "prof rc=$?"
cd $R; ls gpurun_out/r1_prof/*/ | head; cat gpurun_out/r1_bench16.log | cut -c1-300; cat gpurun_out/r1_bench64.log | tail -1 | cut -c1-300
