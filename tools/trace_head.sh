#!/bin/bash
# Phase timing of conv_head_up2r_kernel (s_memtime stamps of waves 0 and 4, -DVSE_TRACE build of conv_head.hip only)
R=$GRAFT_REPO_ROOT; cd $R/video-subtitle-extractor_amd/csrc
OBJS=$(ls build/*.hip.o | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DVSE_TRACE -c conv_head.hip -o build/conv_head.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
(cd $R && python tools/gpu_profile_net.py V4_ch_det ${1:-64} 544 960 --top 3 2>&1 | grep -E "head trace|op 97 " | tail -3)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c conv_head.hip -o build/conv_head.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
