for i in 1 2; do
for v in 0 1; do
DV3_WGRAD_TILE=$v timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('WGRAD_TILE=$v', d['value'], d['ms_per_step'])"
done; done
