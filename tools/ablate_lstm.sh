#!/bin/bash
# What a step of lstm_mfma16_kernel is made of: builds of the library with one part removed each (results are garbage, times are not)
#   here (CPU container):  bash tools/ablate_lstm.sh prepare
#   on the GPU box:        bash tools/ablate_lstm.sh run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/video-subtitle-extractor_amd/csrc
if [ "$1" = prepare ]; then
  mkdir -p $R/build/ab
  for V in BASE NO_LO NO_CELL NO_GX NO_WLOAD WQ4; do
    F=-DLSTM_ABL_$V; [ $V = WQ4 ] && F=-DLSTM16_WQ=4
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $F -c $C/lstm.hip -o $R/build/ab/lstm_$V.o || exit 1
    objs=$(ls $C/build/*.o | grep -v lstm.hip.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/ab/libvse_$V.so $objs $R/build/ab/lstm_$V.o || exit 1
  done
  ls -la $R/build/ab/*.so
else
  for V in BASE WQ4 NO_LO NO_CELL NO_GX NO_WLOAD BASE WQ4; do
    echo -n "$V: "; VSE_LIB_PATH=$R/build/ab/libvse_$V.so python tools/gpu_profile_net.py V2_ch_rec 32 32 768 --ragged --wmin 520 --top 2 2>&1 | grep "kind=12" | head -1 | cut -c1-40
  done
fi
