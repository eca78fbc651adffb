# LDS counters of the wgrad_c8 forms (register-transposing 5063 / transposing-read 5103): SQ_LDS_BANK_CONFLICT (extra LDS cycles),
# SQ_LDS_IDX_ACTIVE (all LDS-array cycles), SQ_LDS_UNALIGNED_STALL, matrix-pipe busy -- per dispatch, --kernel-trace only
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp; export R
cat > /tmp/pmc_wc8.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["R"])
import torch
from deepvoice3_pytorch_amd import ops, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16"); ops.bf16_storage = True
B, C, M, T, J, d = 64, 512, 1024, 804, 3, 3
x8 = ops.to_c8(torch.randn(B, C, T, device=dev)); g8 = ops.to_c8(torch.randn(B, M, T, device=dev))
keep = ops.dropout_keep_c8(B, C, T, 0.05, dev); keep = keep[0] if isinstance(keep, tuple) else keep
for tr in (0, 1, 2):
    L.dv3_debug_set(52, tr)
    for _ in range(3):
        ops.wgrad_gemm_c8(g8, x8, B=B, M=M, Cin=C, T=T, J=J, dil=d, padL=d, n_slabs=8, xmask_c8=keep, drop_scale=1 / 0.95, rows_of_slabs=True)
torch.cuda.synchronize()
PY
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  T=$(echo $C | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wc8_$T -- python /tmp/pmc_wc8.py > $R/gpurun_out/pmc_wc8_$T.log 2>&1; echo "$T rc=$?"
  F=$(ls $R/gpurun_out/pmc_wc8_$T/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$F" ] && python - "$F" <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "wgrad_c8" not in n: continue
    k = (n.split("(")[0][-60:], r["Counter_Name"])
    agg.setdefault(k, []).append(float(r["Counter_Value"]))
for (n, c), v in agg.items():
    print("%-62s %-28s %14.0f  (n=%d)" % (n, c, sum(v) / len(v), len(v)))
PY
done
