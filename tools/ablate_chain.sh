#!/bin/bash
# Timing-only ablations of chain_kernel: rebuild chain.hip with -DVSE_CHAIN_ABL=<mask> and time the mobile detectors' chains
# (tools/chain_check.py --time-only).  usage: ablate_chain.sh "<masks>" [MAXSTAGES]     (mask 0 = the product build; restored at the end)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/video-subtitle-extractor_amd/csrc
for M in $1 0; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DVSE_CHAIN_ABL=$M -c chain.hip -o build/chain.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvse_hip.so $(ls build/*.hip.o | grep -v c3w)
  echo "== VSE_CHAIN_ABL=$M"
  (cd $R && VSE_CHAIN_MAXSTAGES=${2:-2} python tools/chain_check.py --time-only 2>&1 | grep "V4_ch_det_fast\|   op [0-9] chain" | head -7 | cut -c1-110)
done
