#!/bin/bash
# crops of 1 / 2 / 3 consecutive batches recognised together (bench.py --rec-span), alternating on one box
run() { echo -n "$*: "; python bench.py "$@" --no-cpu-baseline --no-roofline --other-mode-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run --rec-span 1
run --rec-span 2
run --rec-span 3 --steps 21
run --rec-span 2 --det-depth 3
done
python bench.py --no-cpu-baseline --other-mode-steps 0 2>&1 >/dev/null | grep per-net
python bench.py --no-cpu-baseline --other-mode-steps 0 --rec-span 1 2>&1 >/dev/null | grep per-net
