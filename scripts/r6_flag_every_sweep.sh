#!/bin/bash
# flag-ordered backward (ABI 43): a signal kernel at every k-th fork point only, against the segments
for K in 2 4 8; do
  export DV3_FLAG_EVERY=$K
  echo "== DV3_FLAG_EVERY=$K"
  timeout 600 python scripts/r6_env_step_ab.py DV3_FLAG_SYNC 0 1 "$@" 2>&1 | grep -v amdgpu.ids
done
