#!/bin/bash
# stream-level knobs of bench.py re-swept on one box with verified-concurrent side streams (round 5): flags, frames/s, ms/step
run() { echo -n "$*: "; python bench.py "$@" --no-cpu-baseline --no-roofline --no-secondary --other-mode-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run
run --det-depth 3
run --det-depth 1
run --rec-streams 3
run --rec-streams 4
run --rec-span 3
run --rec-span 1
run --rec-priority 0
run --det-priority -1 --rec-priority 0
run --det-depth 3 --rec-span 3
run --min-rec-group 0
run --bucket 384
run
