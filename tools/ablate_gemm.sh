#!/bin/bash
# Timing-only ablations of conv_gemm_kernel (results are garbage for ABLATE != 0): rebuild the library with
# -DVSE_ABLATE=n and time single layers (tools/bench_conv.py).  usage: ablate_gemm.sh "<cfgs>" "<layers>"
R=$GRAFT_REPO_ROOT; cd $R/video-subtitle-extractor_amd/csrc
OBJS="build/vse_runtime.hip.o build/conv_mfma.hip.o build/conv_gemm.hip.o build/conv_patch.hip.o build/conv_col.hip.o build/conv_c3.hip.o build/conv_pw.hip.o build/conv_head.hip.o build/conv_stem.hip.o build/simple_ops.hip.o build/prepost.hip.o"
for A in ${ABL:-0 1 2 3 4 5}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DVSE_ABLATE=$A -c conv_gemm.hip -o build/conv_gemm.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
  echo "ABLATE=$A"; (cd $R && python tools/bench_conv.py --cfgs "$1" --layers "$2" 2>&1 | grep -v amdgpu.ids | sed 's/ MISMATCH//g')
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c conv_gemm.hip -o build/conv_gemm.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
