python tools/dw_tile_check.py 2>&1 | grep -v amdgpu.ids | cut -c1-150
python tools/chain_check.py --time 2>&1 | grep -E "chain_check|FAIL|64x544x960" | cut -c1-120
