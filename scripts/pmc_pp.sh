# PMC passes (separate runs, hard timeouts) over the bf16x3 tap-GEMM with the ping-pong main loop
# (auto-picked 128x256 tile, north-star shape)
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp; export R
cat > /tmp/pmc_run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["R"])
sys.argv = ["x"]
import scripts.x3_check as X
X.timeit(0, 1, iters=8)
PY
timeout 55 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_pp_a -- python /tmp/pmc_run.py > $R/gpurun_out/pmc_pp_a.log 2>&1; echo rc=$?
timeout 55 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/pmc_pp_b -- python /tmp/pmc_run.py > $R/gpurun_out/pmc_pp_b.log 2>&1; echo rc=$?
