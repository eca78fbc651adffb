#!/bin/bash
# flag-ordered backward (ABI 43): lag (fork points) of the weight-gradient graph behind the step stream, against the segments
for L in 0 2 4 8; do
  export DV3_FLAG_LAG=$L
  echo "== DV3_FLAG_LAG=$L"
  timeout 600 python scripts/r6_env_step_ab.py DV3_FLAG_SYNC 0 1 "$@" 2>&1 | grep -v amdgpu.ids
done
