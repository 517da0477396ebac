#!/bin/bash
# Timing-only ablations of conv_patch_kernel (results are garbage for ABLATE != 0): rebuild the library with
# -DVSE_ABLATE=n and time the detector's patch-kernel layers (9x9 cin256, final 3x3 + DOT1, 3x3 cin256 -> 64, 7x7).
R=$GRAFT_REPO_ROOT; cd $R/video-subtitle-extractor_amd/csrc
OBJS="build/vse_runtime.hip.o build/conv_mfma.hip.o build/conv_gemm.hip.o build/conv_patch.hip.o build/conv_col.hip.o build/conv_c3.hip.o build/conv_pw.hip.o build/conv_head.hip.o build/conv_stem.hip.o build/simple_ops.hip.o build/prepost.hip.o"
for A in ${ABL:-0 1 2 3 4}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DVSE_ABLATE=$A -c conv_patch.hip -o build/conv_patch.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
  echo "ABLATE=$A"; (cd $R && python tools/gpu_profile_net.py V4_ch_det 16 544 960 --top 70 2>&1 | grep -E "op 65 |op  2 |op  4 |op 96 |op 89 |op 69 " )
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c conv_patch.hip -o build/conv_patch.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
