timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "bf16x3_forward or north or layer" 2>&1 | tail -2
python - <<'PY'
import torch, bench
dev = torch.device("cuda:0")
for mode in ("f16x3", "bf16x3", "f16x3"):
    r = bench.conv_roofline(dev, mode=mode)
    print(mode, r["us_per_launch"], r["frac"], r["variant"])
PY
timeout 600 python -m pytest tests/test_gpu_preset_scale.py -q -x -k "f16x3 and (eval_forward or golden or north_star)" 2>&1 | tail -2
