#!/bin/bash
# kernel trace + stats of the replayed B = 16 step (the preset's own batch size) -> timeline + per-kernel summary
R="${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/r6_b16_tl
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6_b16_tl -- python $R/bench.py --batch 16 --steps 12 --warmup 4 --settle 0 --no-cpu-baseline --no-extras --no-roofline --graph "$@" > $R/gpurun_out/r6_b16_tl.log 2>&1; echo "prof rc=$?"
F=$(ls $R/gpurun_out/r6_b16_tl/*/*kernel_trace.csv | head -1)
S=$(ls $R/gpurun_out/r6_b16_tl/*/*kernel_stats.csv | head -1)
python $R/scripts/r5_timeline.py $F > $R/gpurun_out/r6_b16_timeline_replay.txt 2>&1
cp $S $R/gpurun_out/r6_b16_kernel_stats.csv
cat $R/gpurun_out/r6_b16_timeline_replay.txt | head -70
tail -2 $R/gpurun_out/r6_b16_tl.log | cut -c1-300
rm -rf $R/gpurun_out/r6_b16_tl
