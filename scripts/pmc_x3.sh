# PMC passes over the bf16x3 tap-GEMM at the north-star shape (separate runs per counter group)
R=$PWD; cd /tmp; export TMPDIR=/tmp
cat > /tmp/pmc_run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["R"])
sys.argv = ["x"]
import scripts.x3_check as X
X.timeit(21, 1, iters=5)
PY
export R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_x3_a -- python /tmp/pmc_run.py > $R/gpurun_out/pmc_x3_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/pmc_x3_b -- python /tmp/pmc_run.py > $R/gpurun_out/pmc_x3_b.log 2>&1
cd $R; tail -3 gpurun_out/pmc_x3_a.log gpurun_out/pmc_x3_b.log
