# effective shader clock (GRBM_GUI_ACTIVE / kernel duration) and SQ busy/wait split of the planes tap-GEMM variants:
# full, MFMAs only, no MFMAs -- is the sum-like behaviour a clock (power) effect?
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp; export R
cat > /tmp/pmc_run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["R"])
import torch
from deepvoice3_pytorch_amd import ops
from scripts.planes_ab import x, v, g, bias, B, C, T, k, dev, lib
ops.set_gemm_precision("f16x3")
pk = ops.pack_weights(v, g, glu_cg=C, need_bwd=False)
y = torch.empty(B, C, T, device=dev)
xp = ops.split_planes(x, f16=True)
kw = dict(B=B, Cin=C, Tin=T, M=2 * C, Tout=T, J=k, dil=1, padL=1, mode=ops.EPI_GLU, Cg=C, bias=bias, r=x, residual=1, a_split=pk.fwd_s, y=y, x_planes=xp)
lib.dv3_debug_set(4, 9); lib.dv3_debug_set(7, 0)
for abl in (0, 8, 3):
    lib.dv3_debug_set(6, abl)
    for _ in range(60):
        ops.conv_gemm(x, None, pk.lda, pk.a_half, **kw)
    torch.cuda.synchronize()
PY
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_clock -- python /tmp/pmc_run.py > $R/gpurun_out/pmc_clock.log 2>&1; echo rc=$?
cd $R; python - <<'PY'
import csv, glob, collections
kt = glob.glob("gpurun_out/pmc_clock/*/*kernel_trace.csv")[0]
cc = glob.glob("gpurun_out/pmc_clock/*/*counter_collection.csv")[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    name, d = dur.get(r["Dispatch_Id"], (r["Kernel_Name"], 0))
    if "conv_planes" not in name: continue
    key = name[name.find("conv_planes"):][:60]
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[key]["_dur_ns"].append(d)
for k, v in acc.items():
    n = len(v["_dur_ns"]) // max(1, len([c for c in v if c != "_dur_ns"]))
    tail = lambda xs: sum(xs[len(xs)//2:]) / max(1, len(xs[len(xs)//2:]))
    d = tail(v["_dur_ns"])
    print(k, "dur %.1f us" % (d / 1e3), " ".join("%s=%.3g" % (c, tail(x)) for c, x in v.items() if c != "_dur_ns"),
          "clock %.2f GHz" % (tail(v.get("GRBM_GUI_ACTIVE", [0])) / max(d, 1)))
PY
