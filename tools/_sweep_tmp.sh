for t in 0 1 2; do VSE_DW_TILE=$t python tools/dw_tile_check.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r4_c36_dwtile.log; sort -k2,4 gpurun_out/r4_c36_dwtile.log | cut -c1-150
