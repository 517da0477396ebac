#!/bin/bash
# A/B of a RUN-TIME switch (environment variable $1 = 0 / 1) on one box: per-layer microbench + the end-to-end bench, alternating
V=$1; L=$2
for X in 0 1 0 1; do
  echo "$V=$X"
  env $V=$X python tools/bench_conv.py --cfgs c --layers $L 2>&1 | grep cfgc | cut -c1-80
  env $V=$X python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
done
