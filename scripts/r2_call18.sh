mkdir -p gpurun_out
timeout 800 python scripts/offset_repro.py > gpurun_out/r2q_offset.log 2>&1; echo "rc=$?"; grep "^==" gpurun_out/r2q_offset.log; tail -5 gpurun_out/r2q_offset.log
