#!/bin/bash
# conv_dwpw_kernel with 1 / 2 / 4 pixel tiles of 32 per wave (VSE_DWPW_TPW): the mobile detectors, default (pairs) and layer by layer
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for V in TPW2 TPW1 TPW4 TPW2 TPW1 TPW4; do
  for F in "" "--no-chain"; do
    echo -n "$V $F: "; VSE_LIB_PATH=$R/build/ab/libvse_$V.so python $R/tools/gpu_profile_net.py V4_ch_det_fast 64 544 960 --hilo $F --top 1 2>&1 | grep "total" | cut -c1-80
  done
done
