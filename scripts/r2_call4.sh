python scripts/planes_ab.py 2>&1 | grep -E "f16x3|bf16 d=1 train=0" | cut -c1-330
python scripts/planes_abl.py 2>&1 | tail -4
