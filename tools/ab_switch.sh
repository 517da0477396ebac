#!/bin/bash
# Compiler experiment switches A/B'd on ONE box, alternating (each line: switches, frames/s, ms/step).  They are read only with
# VSE_DEV_BUILD=1 (ir.dev_switch) AND with a development build of the library (engine.load_library refuses the product .so under
# VSE_DEV_BUILD=1: both halves of a switch must see it): `bash tools/build_ab.sh vse_runtime.hip DEV=` first.
# usage: bash tools/ab_switch.sh VSE_TAIL2   (-> VSE_TAIL2=1 / =0 three times)
S=${1:-VSE_TAIL2}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() { echo -n "$1: "; env VSE_DEV_BUILD=1 VSE_LIB_PATH=$R/build/ab/libvse_DEV.so $1 python bench.py --no-cpu-baseline --no-roofline --no-secondary --other-mode-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  run $S=1
  run $S=0
done
