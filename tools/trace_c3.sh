#!/bin/bash
# Phase timing of conv_c3_kernel blocks (s_memtime stamps, -DVSE_TRACE build of conv_c3.hip only): one "[c3 trace]" line per launch.
#   usage: bash tools/trace_c3.sh <layers,comma,separated (tools/bench_conv.py names)>
R=$GRAFT_REPO_ROOT; cd $R/video-subtitle-extractor_amd/csrc
OBJS="build/vse_runtime.hip.o build/conv_mfma.hip.o build/conv_gemm.hip.o build/conv_patch.hip.o build/conv_col.hip.o build/conv_c3.hip.o build/conv_pw.hip.o build/conv_head.hip.o build/conv_stem.hip.o build/simple_ops.hip.o build/prepost.hip.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DVSE_TRACE ${EXTRA} -c conv_c3.hip -o build/conv_c3.hip.o build/conv_pw.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
(cd $R && python tools/bench_conv.py --cfgs c --layers $1 2>&1 | grep -E "c3 trace|cfgc" | sort | uniq -c | sort -rn | head -40)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c conv_c3.hip -o build/conv_c3.hip.o build/conv_pw.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
