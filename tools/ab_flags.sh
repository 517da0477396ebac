#!/bin/bash
# A/B of an extra hipcc flag on the conv kernels: prepare here (builds build/ab/libvse_<tag>.so), run on the GPU box.
#   bash tools/ab_flags.sh prepare <tag> "<flags>"      bash tools/ab_flags.sh run <tag> <layers>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/video-subtitle-extractor_amd/csrc
if [ "$1" = prepare ]; then
  T=$2; F=$3; mkdir -p $R/build/ab/$T
  objs=""
  for f in conv_c3 conv_col conv_gemm conv_patch conv_head; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $F -c $C/$f.hip -o $R/build/ab/$T/$f.o || exit 1
    objs="$objs $R/build/ab/$T/$f.o"
  done
  rest=$(ls $C/build/*.o | grep -v -E "conv_c3|conv_col|conv_gemm|conv_patch|conv_head")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/ab/libvse_$T.so $rest $objs && ls -la $R/build/ab/libvse_$T.so
else
  T=$2; L=$3
  for i in 1 2; do
    echo "== base"; python tools/bench_conv.py --cfgs ${4:-c} --layers $L 2>&1 | grep cfg | cut -c1-70
    echo "== $T"; VSE_LIB_PATH=$R/build/ab/libvse_$T.so python tools/bench_conv.py --cfgs ${4:-c} --layers $L 2>&1 | grep cfg | cut -c1-70
  done
fi
