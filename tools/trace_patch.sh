#!/bin/bash
# Phase timing of conv_patch_kernel blocks (s_memtime stamps, -DVSE_TRACE build of conv_patch.hip only).  Prints one "[patch trace]" line per patch-kernel launch of the detector.
R=$GRAFT_REPO_ROOT; cd $R/video-subtitle-extractor_amd/csrc
OBJS="build/vse_runtime.hip.o build/conv_mfma.hip.o build/conv_gemm.hip.o build/conv_patch.hip.o build/conv_col.hip.o build/conv_c3.hip.o build/conv_pw.hip.o build/conv_head.hip.o build/conv_stem.hip.o build/simple_ops.hip.o build/prepost.hip.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DVSE_TRACE -c conv_patch.hip -o build/conv_patch.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
(cd $R && python tools/gpu_profile_net.py V4_ch_det ${1:-64} 544 960 --top 5 2>&1 | grep "patch trace" | sort | uniq -c | sort -rn | head -40)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c conv_patch.hip -o build/conv_patch.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
