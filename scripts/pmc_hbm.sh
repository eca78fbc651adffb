# HBM traffic of the bf16x3 tap-GEMM (north-star shape), one derived counter per pass, each under a hard timeout
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp; export R
cat > /tmp/pmc_run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["R"])
sys.argv = ["x"]
import torch
import scripts.x3_check as X
from deepvoice3_pytorch_amd import ops
X.timeit(21, 1, iters=5)
a = torch.randn(64 * 1024 * 1024, device="cuda")   # calibration: axpby reads 256 MiB, writes 256 MiB (4 B per lane)
for _ in range(3):
    ops.axpby(a, None, 2.0)
torch.cuda.synchronize()
PY
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_x3_$C -- python /tmp/pmc_run.py > $R/gpurun_out/pmc_x3_$C.log 2>&1; echo "$C rc=$?"
done
