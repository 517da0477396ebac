#!/bin/bash
# ragged grouping cost model (pipeline._ragged_partition): floor / launch cost sweep on one box, alternating
run() { echo -n "$*: "; python bench.py "$@" --no-cpu-baseline --no-roofline --other-mode-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run --ragged-floor 6000 --ragged-launch-cost 5000
run --ragged-floor 6000 --ragged-launch-cost 2500
run --ragged-floor 3000 --ragged-launch-cost 1200
run --ragged-floor 10000 --ragged-launch-cost 8000
run --rec-mode bucketed
done
