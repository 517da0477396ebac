#!/bin/bash
# detector stream modes A/B'd on one box (each line: mode, frames/s, ms/step): own (torch pool streams), shared (one stream), independent (verified)
run() { echo -n "$*: "; python bench.py "$@" --no-cpu-baseline --no-roofline --no-secondary --other-mode-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  run --det-stream-mode own
  run --det-stream-mode shared
  run --det-stream-mode independent
done
run --det-stream-mode shared --det-depth 3
run --det-stream-mode shared --det-depth 1
run --det-stream-mode own --det-depth 1
run --det-stream-mode shared --rec-streams 1
run --det-stream-mode shared --rec-streams 3
run --det-stream-mode shared --rec-priority 0
