#!/bin/bash
# Timing-only ablations / variants of conv_c3w_kernel: rebuild conv_c3w.hip with the given -D flags and time single layers
# (tools/bench_conv.py, VSE_C3_WIDE=1).  usage: ablate_c3w.sh "<layers>" "<flags>" "<flags>" ...   ("-" = the default build)
R=$GRAFT_REPO_ROOT; cd $R/video-subtitle-extractor_amd/csrc
OBJS=$(ls build/*.hip.o | tr '\n' ' ')
L=$1; shift
for F in "$@"; do
  FF=$F; [ "$F" = "-" ] && FF=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $FF -c conv_c3w.hip -o build/conv_c3w.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
  echo "== $F"; (cd $R && VSE_C3_WIDE=1 python tools/bench_conv.py --cfgs c --layers "$L" 2>&1 | grep -v amdgpu.ids | sed 's/ MISMATCH//g')
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c conv_c3w.hip -o build/conv_c3w.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $OBJS
echo "== c3 (VSE_C3_WIDE unset)"; (cd $R && python tools/bench_conv.py --cfgs c --layers "$L" 2>&1 | grep -v amdgpu.ids)
