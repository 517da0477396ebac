#!/bin/bash
# A/B timing of a compile-time switch of one kernel file on ONE box (boxes differ by several %): builds the library with
# -D<MACRO>=0 and =1 (twice, alternating) and runs the per-layer microbench on each.
#   usage: bash tools/ab.sh <conv_gemm|conv_patch|conv_head> <MACRO> <cfg: d|p> <layers,comma,separated>
R=$GRAFT_REPO_ROOT; cd $R/video-subtitle-extractor_amd/csrc
OBJS=$(ls build/*.hip.o | tr "\n" " ")
F=$1; M=$2; C=$3; L=$4
for V in ${VALS:-0 1 0 1}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -D$M=$V -c $F.hip -o build/$F.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
  export $M=$V       # (switches mirrored by the graph compiler read the same name from the environment)
  echo "$M=$V"; (cd $R && python tools/bench_conv.py --cfgs $C --layers $L 2>&1 | grep -v amdgpu.ids | cut -c1-80)
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c $F.hip -o build/$F.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
