python - <<'PY'
import torch, bench
dev = torch.device("cuda:0")
for _ in range(2):
    r = bench.wgrad_roofline(dev)
    print("wgrad", r["us_per_launch"], r["achieved"], r["frac"])
PY
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "wgrad or layer" 2>&1 | tail -2
bash scripts/pmc_hbm_r2.sh 2>&1 | grep -E "hbm_bytes_per_launch|read_bytes|kernel"
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-200
