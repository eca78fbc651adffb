# PMC passes over the bf16x3 tap-GEMM at the north-star shape (separate runs per counter group; no trace domains
# beyond --kernel-trace), plus an HBM-traffic pass calibrated on a streaming kernel with a known byte count.
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp; export R
cat > /tmp/pmc_run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["R"])
sys.argv = ["x"]
import torch
import scripts.x3_check as X
from deepvoice3_pytorch_amd import ops
X.timeit(21, 1, iters=5)
# calibration: axpby over 64M floats reads 256 MiB and writes 256 MiB with 4-byte-per-lane accesses
a = torch.randn(64 * 1024 * 1024, device="cuda")
for _ in range(3):
    ops.axpby(a, None, 2.0)
torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_x3_a -- python /tmp/pmc_run.py > $R/gpurun_out/pmc_x3_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/pmc_x3_b -- python /tmp/pmc_run.py > $R/gpurun_out/pmc_x3_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_x3_hbm -- python /tmp/pmc_run.py > $R/gpurun_out/pmc_x3_hbm.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_x3_clk -- python /tmp/pmc_run.py > $R/gpurun_out/pmc_x3_clk.log 2>&1
cd $R; ls gpurun_out/pmc_x3_hbm/*/ 
