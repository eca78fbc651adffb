# PMC passes (one counter group per pass, --kernel-trace only: gpurun refuses --pmc with the trace domains) of the kernels
# bench.py's rooflines time at the north-star shape: forward tap-GEMM (256 x 256 k16 ping-pong, variant 5101), the
# all-taps wgrad (3030) and the c8 planes tap-GEMM (8090): HBM bytes (FETCH_SIZE / WRITE_SIZE calibrated on a known copy
# as MI355X_MICROARCH.md prescribes) and matrix-pipe busy cycles.  Writes gpurun_out/r06_hbm_traffic.json + r06_mfma_busy.json
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp; export R; export DV3_BENCH_EAGER_TIMING=1
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
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE"; do
  T=$(echo $C | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r6_$T -- python /tmp/pmc_run.py > $R/gpurun_out/pmc_r6_$T.log 2>&1; echo "$T rc=$?"
done
cd $R; python - <<'PY'
import csv, glob, json, os, re
def newest(pat):
    return sorted(glob.glob(pat), key=os.path.getmtime)[-1]
def vals(tag, counter):
    f = newest("gpurun_out/pmc_r6_%s/*/*counter_collection.csv" % tag)
    out = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out
def durs(tag):
    f = newest("gpurun_out/pmc_r6_%s/*/*kernel_trace.csv" % tag)
    out = {}
    for r in csv.DictReader(open(f)):
        out.setdefault(r["Kernel_Name"], []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return out
def pick(d, key):
    ks = [k for k in d if key in k]
    k = max(ks, key=lambda k: len(d[k]))
    v = d[k]
    return k, sum(v[-5:]) / len(v[-5:])       # the timed launches (warm)
F, W = vals("FETCH_SIZE", "FETCH_SIZE"), vals("WRITE_SIZE", "WRITE_SIZE")
_, cf = pick(F, "axpby"); _, cw = pick(W, "axpby")
fcal, wcal = 262144.0 / cf, 262144.0 / cw
log = open("gpurun_out/pmc_r6_FETCH_SIZE.log").read()
var = dict(re.findall(r"VARIANT (\w+) (\d+)", log))
res, busy = {}, {}
names = (("conv_fwd:%s" % var.get("conv"), "conv_gemm_pp2_kernel" if var.get("conv", "").endswith("101") else "conv_gemm_bf16x3_kernel", 135792640),
         ("wgrad:%s" % var.get("wgrad"), "wgrad_taps2_kernel" if 40 <= int(var.get("wgrad", "0")) % 100 <= 47 else "wgrad_taps_kernel", 202899456), ("conv_fwd:%s" % var.get("convc8"), "conv_c8pp_kernel" if var.get("convc8", "").startswith("9") else "conv_planes_kernel", 67897344))
D = durs("SQ_VALU_MFMA_BUSY_CYCLES")
for key, name, alg in names:
    kn, f = pick(F, name); _, w = pick(W, name)
    rd, wr = f * fcal * 1024, w * wcal * 1024
    res[key] = dict(kernel=kn[:140], fetch_size_kb_raw=f, write_size_kb_raw=w, fetch_calibration=round(fcal, 4), write_calibration=round(wcal, 4),
                    read_bytes=int(rd), write_bytes=int(wr), hbm_bytes_per_launch=int(rd + wr), algorithmic_bytes=alg,
                    source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/pmc_r6.sh), round 6; calibrated on dv3 axpby over 64 Mi floats")
    b = {}
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
        try:
            b[c] = pick(vals("SQ_VALU_MFMA_BUSY_CYCLES", c), name)[1]
        except Exception:
            b[c] = None
    try:
        b["GRBM_GUI_ACTIVE"] = pick(vals("GRBM_GUI_ACTIVE", "GRBM_GUI_ACTIVE"), name)[1]
    except Exception:
        b["GRBM_GUI_ACTIVE"] = None
    b["duration_us_in_pmc_pass"] = pick(D, name)[1] / 1e3
    # matrix-pipe busy: MFMA busy cycles are summed over the 1024 SIMDs (= 32 cycles x MFMAs issued), SQ_BUSY_CYCLES
    # over the 32 shader engines; GRBM_GUI_ACTIVE is not a usable cycle base on this stack (it implies > 2.4 GHz)
    if b.get("SQ_VALU_MFMA_BUSY_CYCLES") and b.get("SQ_BUSY_CYCLES"):
        cyc = b["SQ_BUSY_CYCLES"] / 32.0
        b["launch_cycles_from_SQ_BUSY_CYCLES_over_32_SEs"] = int(cyc)
        b["implied_clock_ghz"] = round(cyc / (b["duration_us_in_pmc_pass"] * 1e3), 3)
        b["mfma_busy_cycles_per_simd"] = int(b["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0)
        b["mfma_busy_frac_of_launch_cycles"] = round(b["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, 4)
    busy[key] = b
json.dump(res, open("gpurun_out/r06_hbm_traffic.json", "w"), indent=1)
json.dump(busy, open("gpurun_out/r06_mfma_busy.json", "w"), indent=1)
print(json.dumps(res, indent=1)); print(json.dumps(busy, indent=1))
PY
