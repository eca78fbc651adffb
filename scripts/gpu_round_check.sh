# tests + bench (both batches, both GEMM modes, synthesis) + rocprofv3 kernel stats; outputs under gpurun_out/
R=$PWD; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r1_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r1_tests.log
python bench.py > gpurun_out/r1_bench64.log 2>gpurun_out/r1_bench64.err; echo "b64 rc=$?"; tail -c 400 gpurun_out/r1_bench64.err
python bench.py --batch 16 --no-cpu-baseline --no-roofline > gpurun_out/r1_bench16.log 2>&1; echo "b16 rc=$?"
python bench.py --gemm f32 --steps 10 --no-cpu-baseline --no-roofline > gpurun_out/r1_bench64_f32.log 2>&1; echo "b64 f32 rc=$?"
python bench.py --mode synth > gpurun_out/r1_synth.log 2>&1; echo "synth rc=$?"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1_prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r1_prof.log 2>&1; echo "prof rc=$?"
cd $R; for f in gpurun_out/r1_bench64.log gpurun_out/r1_bench16.log gpurun_out/r1_bench64_f32.log gpurun_out/r1_synth.log; do tail -1 $f | cut -c1-330; done
