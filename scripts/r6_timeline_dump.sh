#!/bin/bash
# kernel trace of the replayed train step, every launch of the backward window listed (scripts/r5_timeline.py ... dump)
R="${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/r6_tl
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r6_tl -- python $R/bench.py --steps 6 --warmup 3 --settle 0 --no-cpu-baseline --no-extras --no-roofline --graph "$@" > $R/gpurun_out/r6_tl.log 2>&1; echo "prof rc=$?"
F=$(ls $R/gpurun_out/r6_tl/*/*kernel_trace.csv | head -1)
python $R/scripts/r5_timeline.py $F ${DUMP:-dump} > $R/gpurun_out/r6_timeline_dump.txt 2>&1
rm -rf $R/gpurun_out/r6_tl
