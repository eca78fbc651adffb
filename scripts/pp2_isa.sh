#!/bin/bash
# ISA of ONE instantiation of the 256 x 256 tap-GEMM (f16x3): scripts/pp2_isa.sh <ORD> [MASK=false] -> /tmp/isa/pp2_<ORD>_<MASK>.s
# prints registers / spills and the scratch accesses with their line numbers (none may sit inside the main loop)
set -e
ORD=${1:-0}; MASK=${2:-false}
mkdir -p /tmp/isa
cd "$(dirname "$0")/../deepvoice3_pytorch_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDV3_EXPERIMENTS -DDV3_PP2_ISA_ONLY -DDV3_PP2_ISA_ORD=$ORD -DDV3_PP2_ISA_MASK=$MASK \
  -S --cuda-device-only conv_gemm_pp2.hip -o /tmp/isa/pp2_${ORD}_${MASK}.s 2>&1 | grep -v hip-link || true
grep -n "\.vgpr_count\|\.vgpr_spill\|\.sgpr_spill\|private_segment_fixed" /tmp/isa/pp2_${ORD}_${MASK}.s
grep -n "scratch_\|Loop Header\|s_endpgm" /tmp/isa/pp2_${ORD}_${MASK}.s | head -40
