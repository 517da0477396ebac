#!/bin/bash
# Whole-pipeline A/B of two git revisions on ONE GPU box (boxes differ by several %).
#   here (CPU container):  bash tools/ab_bench.sh prepare <old-rev>     # extracts + builds build/ab/old (travels with gpurun)
#   on the GPU box:        bash tools/ab_bench.sh run                   # alternates bench.py of build/ab/old and of the tree
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "$1" = prepare ]; then
  rm -rf $R/build/ab/old && mkdir -p $R/build/ab/old
  (cd $R && git archive $2 | tar -x -C build/ab/old)
  (cd $R/build/ab/old && python __graft_entry__.py | tail -1)
else
  for i in 1 2 3; do
    for T in build/ab/old .; do
      (cd $R/$T && python bench.py --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$T', d['value'], d['ms_per_step'])")
    done
  done
fi
