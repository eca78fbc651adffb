mkdir -p gpurun_out
timeout 700 python scripts/c8_check.py > gpurun_out/r2s_c8.log 2>&1; echo "c8 rc=$?"; grep -v "^$" gpurun_out/r2s_c8.log | tail -60
timeout 600 python scripts/offset_repro.py > gpurun_out/r2r_offset.log 2>&1; echo "rc=$?"; grep "^==" gpurun_out/r2r_offset.log | cut -c1-300; tail -3 gpurun_out/r2r_offset.log
