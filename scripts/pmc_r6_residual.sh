# Where the forward tap-GEMM's surplus HBM reads come from: the same launch with the residual (r = x, the Conv1dGLU form) and
# without it (residual = 0) -- FETCH_SIZE per dispatch (x2 calibration as scripts/pmc_r6.sh), --kernel-trace only
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp; export R; export DV3_BENCH_EAGER_TIMING=1
cat > /tmp/pmc_res.py <<'PY'
import sys, os, math
sys.path.insert(0, os.environ["R"])
import torch
from deepvoice3_pytorch_amd import ops, _lib
dev = torch.device("cuda:0")
B, C, T, k = 64, 256, 1024, 3
torch.manual_seed(0)
x = torch.randn(B, C, T, device=dev)
v = torch.randn(2 * C, C, k, device=dev) * math.sqrt(4.0 * 0.95 / (k * C))
g = v.reshape(2 * C, -1).norm(dim=1).view(-1, 1, 1).clone()
bias = torch.zeros(2 * C, device=dev)
pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
y = torch.empty(B, C, T, device=dev)
other = torch.randn(B, C, T, device=dev)
for tag, r, res in (("residual r = x", x, 1), ("no residual", None, 0), ("residual from another tensor", other, 1)):
    for _ in range(3):
        ops.conv_gemm(x, pk.fwd, pk.lda, pk.a_half, B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C,
                      bias=bias, r=r, residual=res, y=y, a_split=pk.fwd_s)
    torch.cuda.synchronize()
    print("ORDER", tag, _lib.lib().dv3_debug_get(10))
a = torch.randn(64 * 1024 * 1024, device="cuda")
for _ in range(3):
    ops.axpby(a, None, 2.0)
torch.cuda.synchronize()
PY
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_res -- python /tmp/pmc_res.py > $R/gpurun_out/pmc_res.log 2>&1; echo "rc=$?"
grep ORDER $R/gpurun_out/pmc_res.log
python - <<'PY'
import csv, glob, os
f = sorted(glob.glob(os.environ["R"] + "/gpurun_out/pmc_res/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
ax = [float(r["Counter_Value"]) for r in rows if "axpby" in r["Kernel_Name"]]
cal = 256.0 * 1024 / (sum(ax) / len(ax))          # axpby reads 256 MiB: KB reported -> calibration
conv = [float(r["Counter_Value"]) for r in rows if "conv_gemm_pp2" in r["Kernel_Name"]]
print("calibration %.4f; conv_gemm_pp2 dispatches in issue order (MB read):" % cal, ["%.1f" % (v * cal * 1024 / 1e6) for v in conv])
PY
