#!/bin/bash
# lstm_mfma16_kernel with 2 vs 4 weight fragments in flight (LSTM16_WQ; 4 spills 15 VGPRs at the 128-register budget): V2_ch_rec per-op time
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for V in WQ4 WQ2 WQ4 WQ2; do
  echo -n "$V: "; VSE_LIB_PATH=$R/build/ab/libvse_$V.so python $R/tools/gpu_profile_net.py V2_ch_rec 32 32 768 --ragged --wmin 520 --top 3 2>&1 | grep "total\|kind=12" | cut -c1-150 | tr '\n' ' '; echo
done
