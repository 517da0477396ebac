#!/bin/bash
# A/B of a conv_stem.hip compile-time switch with tools/prof_stem.py (detector forward + the stem op, both models), alternating builds
R=$GRAFT_REPO_ROOT; cd $R/video-subtitle-extractor_amd/csrc
OBJS=$(ls build/*.hip.o | tr "\n" " ")
M=${1:-VSE_STEM_WIDE}
for V in 0 1 0 1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -D$M=$V -c conv_stem.hip -o build/conv_stem.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
  echo "$M=$V"; (cd $R && python tools/prof_stem.py 2>&1 | grep fused)
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c conv_stem.hip -o build/conv_stem.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
