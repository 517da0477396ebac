#!/bin/bash
# the other configurations of BASELINE.json + the sequential form, same build, one box (secondary figures of profiles/README.md)
run() { echo -n "$*: "; python bench.py "$@" --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run
run --no-overlap
run --det-depth 1
run --height 2160 --width 3840 --batch 32
run --height 720 --width 1280 --models fast
run --models fast-real
run --models v2
run --height 2160 --width 3840 --batch 8 --limit-side 3840
