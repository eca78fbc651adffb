timeout 900 python scripts/tile_rel_ab.py 2>&1 | grep -v amdgpu.ids | tail -8
