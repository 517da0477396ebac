#!/bin/bash
# usage: sweep_bench.sh "<flag>" v1 v2 ...   -> one line per value: flag value frames/s ms_per_step
F=$1; shift
for v in "$@"; do
  python bench.py --no-cpu-baseline --no-roofline --no-secondary $F $v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$F', '$v', d['value'], d['ms_per_step'])"
done
