#!/bin/bash
# A/B timing of a compile-time switch of conv_gemm.hip on ONE box (boxes differ by several %): builds the library with
# -D$1=0 and -D$1=1 and runs the per-layer microbench on both.   usage: bash tools/ab_gemm.sh VSE_GEMM_ASM [layers]
R=$GRAFT_REPO_ROOT; cd $R/video-subtitle-extractor_amd/csrc
OBJS="build/vse_runtime.hip.o build/conv_mfma.hip.o build/conv_gemm.hip.o build/conv_patch.hip.o build/conv_head.hip.o build/simple_ops.hip.o build/prepost.hip.o"
L=${2:-det_1x1_896_256,det_1x1_1216_512,det_1x1_1920_768,det_3x3_192_192,det_3x3_160_160,det_3x3_256_160,det_3x3_768_192,rec_3x3_192_192,rec_1x1_1920_768,det_3x3_128_128}
for V in 0 1 0 1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -D$1=$V -c conv_gemm.hip -o build/conv_gemm.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
  echo "$1=$V"; (cd $R && python tools/bench_conv.py --cfgs d --layers $L 2>&1 | grep -v amdgpu.ids | cut -c1-80)
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c conv_gemm.hip -o build/conv_gemm.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
