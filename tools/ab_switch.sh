#!/bin/bash
# Compiler experiment switches A/B'd on ONE box, alternating (each line: switches, frames/s, ms/step).  They are read only with
# VSE_DEV_BUILD=1 (ir.dev_switch); the product .so is used as is.  usage: bash tools/ab_switch.sh VSE_TAIL2   (-> VSE_TAIL2=1 / =0 three times)
S=${1:-VSE_TAIL2}
run() { echo -n "$1: "; env VSE_DEV_BUILD=1 $1 python bench.py --no-cpu-baseline --no-roofline --no-secondary --other-mode-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  run $S=1
  run $S=0
done
