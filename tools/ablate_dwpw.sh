#!/bin/bash
# Development build only (VSE_DEV_BUILD=1 python __graft_entry__.py --force): what bounds conv_dwpw_kernel?  Per-op times of the mobile detector's
# fused depthwise + 1x1 layers with parts of the kernel switched off (VSE_DWPW_ABL: 1 no stores, 2 middle filter row only, 4 all rows loaded
# but only the middle one multiplied, 8 no MFMAs).
for hl in "--hilo" "--hilo --no-chain"; do
for a in 0 1 2 4 8 3 15; do
  echo "== $hl VSE_DWPW_ABL=$a"
  VSE_DWPW_ABL=$a python tools/gpu_profile_net.py V4_ch_det_fast 64 544 960 $hl --top 80 2>&1 | grep -E "conv_dwpw|total" | cut -c1-150
done; done
