#!/bin/bash
# stream-level knobs of bench.py A/B'd on one box (each line: flags, frames/s, ms/step)
run() { echo -n "$*: "; python bench.py "$@" --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run --det-depth 2
run --det-depth 2 --rec-streams 1
run --det-depth 2 --rec-streams 3
run --det-depth 2 --rec-streams 4
run --det-depth 2 --rec-priority 0
run --det-depth 2 --det-priority -1 --rec-priority 0
run --det-depth 2 --min-rec-group 0
run --det-depth 2 --bucket 128
run --det-depth 2 --bucket 512
run --det-depth 2
