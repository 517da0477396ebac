#!/bin/bash
# A/B of a conv_common.h switch: every conv kernel file is rebuilt with -D$1=0 / =1
R=$GRAFT_REPO_ROOT; cd $R/video-subtitle-extractor_amd/csrc
M=$1; L=$2
FILES="conv_mfma conv_gemm conv_patch conv_col conv_c3 conv_pw conv_head conv_stem"
OBJS="build/vse_runtime.hip.o build/conv_mfma.hip.o build/conv_gemm.hip.o build/conv_patch.hip.o build/conv_col.hip.o build/conv_c3.hip.o build/conv_pw.hip.o build/conv_head.hip.o build/conv_stem.hip.o build/simple_ops.hip.o build/prepost.hip.o"
for V in 0 1 0 1; do
  for F in $FILES; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -D$M=$V -c $F.hip -o build/$F.hip.o 2>/dev/null & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
  echo "$M=$V"; (cd $R && python tools/bench_conv.py --cfgs c --layers $L 2>&1 | grep -v amdgpu.ids | cut -c1-80)
  (cd $R && python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])")
done
for F in $FILES; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c $F.hip -o build/$F.hip.o 2>/dev/null & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
