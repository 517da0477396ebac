#!/bin/bash
# Phase timing of chain_kernel tiles (s_memtime stamps of one block, -DVSE_CHAIN_TRACE build of chain.hip only): one "[chain trace]" line per launch.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/video-subtitle-extractor_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DVSE_CHAIN_TRACE -c chain.hip -o build/chain.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $(ls build/*.hip.o | grep -v c3w)
(cd $R && python tools/gpu_profile_net.py ${1:-V4_ch_det_fast} ${2:-64} 544 960 --hilo --top 3 2>&1 | grep "chain trace" | tail -${3:-12})
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c chain.hip -o build/chain.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $(ls build/*.hip.o | grep -v c3w)
