#!/bin/bash
# crops of 1 / 2 / 3 / 4 consecutive batches recognised together (bench.py --rec-span), alternating on one box
run() { echo -n "$*: "; python bench.py "$@" --no-cpu-baseline --no-roofline --other-mode-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run --rec-span 2 --steps 24
run --rec-span 3 --steps 24
run --rec-span 4 --steps 24
run --rec-span 3 --steps 24 --rec-streams 3
done
