#!/bin/bash
# Recogniser GPU time per step (sequential profiled pass, bench stderr) and end-to-end frames/s against the grouping's launch cost
# and the number of batches recognised together.  usage: ab_rec_groups.sh "<launch costs>" "<spans>"
R=$GRAFT_REPO_ROOT; cd $R
for L in ${1:-5000 30000 1000000}; do for S in ${2:-2 3}; do
  python bench.py --no-cpu-baseline --other-mode-steps 0 --ragged-launch-cost $L --rec-span $S 2> /tmp/ab.err > /tmp/ab.json
  echo "launch-cost $L span $S: $(python -c "import json;d=json.load(open('/tmp/ab.json'));print(d['value'],'frames/s')") | $(grep 'per-net GPU ms' /tmp/ab.err | sed 's/.*-> per step: //')"
done; done
