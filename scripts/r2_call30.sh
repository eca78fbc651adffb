timeout 600 python -m pytest tests/test_gpu_c8.py -m gpu -q -k "converters" 2>&1 | tail -2
timeout 900 python scripts/tile_thr_ab.py 2>&1 | grep -v amdgpu.ids | tail -12
