#!/bin/bash
# detector working set vs the 256 MiB Infinity Cache: frames per step 64 / 16 / 8 / 4 at the same total number of frames
run() { echo -n "$*: "; python bench.py "$@" --no-cpu-baseline --no-roofline --other-mode-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run --batch 64 --steps 20
run --batch 16 --steps 80 --warmup 8
run --batch 8 --steps 160 --warmup 8 --det-depth 3
run --batch 4 --steps 320 --warmup 8 --det-depth 4
run --batch 64 --steps 20
