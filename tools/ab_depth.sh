#!/bin/bash
# stream-level / grouping knobs of bench.py A/B'd on one box (each line: flags, frames/s, ms/step)
run() { echo -n "$*: "; python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['workload'][-60:])"; }
run
run --bucket 192
run --bucket 320
run --bucket 384
run --batch-round 2
run --batch-round 8
run
