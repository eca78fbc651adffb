#!/bin/bash
# ISA of ONE instantiation of the 128 x 64 split tap-GEMM tile: scripts/x3_isa.sh <DPJ 0|1|3> [MASK=false] [F16=true] -> /tmp/isa/x3_<DPJ>_<MASK>_<F16>.s
# prints registers / spills; read the main loop's s_waitcnt vmcnt(N) against the loads in flight
set -e
DPJ=${1:-1}; MASK=${2:-false}; F16=${3:-true}
mkdir -p /tmp/isa
cd "$(dirname "$0")/../deepvoice3_pytorch_amd/csrc"
OUT=/tmp/isa/x3_${DPJ}_${MASK}_${F16}.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDV3_X3_ISA_ONLY -DDV3_X3_ISA_DPJ=$DPJ -DDV3_X3_ISA_MASK=$MASK -DDV3_X3_ISA_F16=$F16 \
  -S --cuda-device-only conv_gemm_bf16x3.hip -o $OUT 2>&1 | grep -i "error" -A5 || true
grep -n "\.vgpr_count\|\.vgpr_spill\|private_segment_fixed" $OUT
grep -n "Loop Header" $OUT
