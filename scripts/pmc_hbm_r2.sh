# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, calibrated on a known copy as MI355X_MICROARCH.md prescribes)
# of the kernels bench.py's rooflines time at the north-star shape: forward tap-GEMM (variant 5091), wgrad (3030, all taps)
# and the single-term planes tap-GEMM on c8 tensors (8090).
# Writes profiles-ready JSON to gpurun_out/r02_hbm_traffic.json.
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp; export R
cat > /tmp/pmc_run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["R"])
import torch
import bench
from deepvoice3_pytorch_amd import ops
dev = torch.device("cuda:0")
print("VARIANT conv", bench.conv_roofline(dev, iters=5)["variant"])
print("VARIANT wgrad", bench.wgrad_roofline(dev, iters=5)["variant"])
print("VARIANT convc8", bench.conv_roofline(dev, iters=5, c8=True)["variant"])
a = torch.randn(64 * 1024 * 1024, device="cuda")   # calibration: axpby reads 256 MiB, writes 256 MiB
for _ in range(3):
    ops.axpby(a, None, 2.0)
torch.cuda.synchronize()
PY
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r2_$C -- python /tmp/pmc_run.py > $R/gpurun_out/pmc_r2_$C.log 2>&1; echo "$C rc=$?"
done
cd $R; python - <<'PY'
import csv, glob, json, re
def last_vals(counter):
    f = glob.glob("gpurun_out/pmc_r2_%s/*/*counter_collection.csv" % counter)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out
F, W = last_vals("FETCH_SIZE"), last_vals("WRITE_SIZE")
def pick(d, key):
    ks = [k for k in d if key in k]
    k = max(ks, key=lambda k: len(d[k]))
    v = d[k]
    return k, sum(v[-5:]) / len(v[-5:])       # the timed launches (warm)
_, cf = pick(F, "axpby"); _, cw = pick(W, "axpby")
fcal, wcal = 262144.0 / cf, 262144.0 / cw
log = open("gpurun_out/pmc_r2_FETCH_SIZE.log").read()
var = dict(re.findall(r"VARIANT (\w+) (\d+)", log))
res = {}
wname = "wgrad_taps_kernel" if var.get("wgrad", "").endswith("30") else "wgrad_gemm_bf16x3_kernel"
for key, name, alg in (("conv_fwd:%s" % var.get("conv"), "conv_gemm_bf16x3_kernel", 135792640), ("wgrad:%s" % var.get("wgrad"), wname, None),
                       ("conv_fwd:%s" % var.get("convc8"), "conv_planes_kernel", 67897344)):
    kn, f = pick(F, name); _, w = pick(W, name)
    rd, wr = f * fcal * 1024, w * wcal * 1024
    res[key] = dict(kernel=kn[:140], fetch_size_kb_raw=f, write_size_kb_raw=w, fetch_calibration=round(fcal, 4), write_calibration=round(wcal, 4),
                    read_bytes=int(rd), write_bytes=int(wr), hbm_bytes_per_launch=int(rd + wr), algorithmic_bytes=alg,
                    source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/pmc_hbm_r2.sh), round 2; calibrated on dv3 axpby over 64 Mi floats")
json.dump(res, open("gpurun_out/r02_hbm_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
