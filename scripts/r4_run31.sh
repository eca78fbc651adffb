#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
bash scripts/r4_prof.sh r04c_vctk --preset deepvoice3_vctk --gemm bf16 --no-graph 2>&1 | grep -i "spk\|total kernel"
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r04c_vctk_prof/runc/*_kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'spk_' in r['Kernel_Name']]
for r in rows[-16:]:
    print(r['Kernel_Name'][22:44], r['Grid_Size_X'], r['Grid_Size_Z'], r['VGPR_Count'], r['LDS_Block_Size'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
PY
