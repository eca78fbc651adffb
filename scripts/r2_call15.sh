mkdir -p gpurun_out
timeout 600 python scripts/poison_check.py > gpurun_out/r2n_poison.log 2>&1; echo "poison rc=$?"; tail -60 gpurun_out/r2n_poison.log
