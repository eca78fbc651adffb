#!/bin/bash
# kernel trace of the eager (two real streams) and of the replayed train step -> scripts/r5_timeline.py
R="${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for MODE in no-graph graph; do
  rm -rf $R/gpurun_out/r5_tl_$MODE
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5_tl_$MODE -- python $R/bench.py --steps 6 --warmup 3 --settle 0 --no-cpu-baseline --no-extras --no-roofline --$MODE "$@" > $R/gpurun_out/r5_tl_$MODE.log 2>&1; echo "prof rc=$?"
  F=$(ls $R/gpurun_out/r5_tl_$MODE/*/*kernel_trace.csv | head -1)
  python $R/scripts/r5_timeline.py $F > $R/gpurun_out/r5_timeline_$MODE.txt 2>&1
  echo "== $MODE"; cat $R/gpurun_out/r5_timeline_$MODE.txt
  rm -rf $R/gpurun_out/r5_tl_$MODE
done
