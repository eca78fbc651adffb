#!/bin/bash
# round 4: full GPU test tier of the final tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r36_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r36_tests.log
tail -4 gpurun_out/r36_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r36_smoke.log 2>&1; tail -2 gpurun_out/r36_smoke.log
