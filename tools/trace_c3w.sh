#!/bin/bash
# Phase timing of conv_c3w_kernel blocks (s_memtime stamps, -DVSE_TRACE build of conv_c3w.hip only): one "[c3w trace]" line per launch.
R=$GRAFT_REPO_ROOT; cd $R/video-subtitle-extractor_amd/csrc
OBJS=$(ls build/*.hip.o | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DVSE_TRACE -c conv_c3w.hip -o build/conv_c3w.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
(cd $R && VSE_C3_WIDE=1 python tools/gpu_profile_net.py V4_ch_det ${1:-64} 544 960 --top 5 2>&1 | grep "c3w trace" | tail -8)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c conv_c3w.hip -o build/conv_c3w.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
