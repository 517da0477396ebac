run() { echo "== $*"; python bench.py "$@" --no-cpu-baseline --other-mode-steps 0 2>/tmp/err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; grep per-net /tmp/err | cut -c1-260; }
for i in 1 2; do
run
run --ragged-launch-cost 18000 --ragged-floor 12000
run --ragged-launch-cost 12000 --ragged-floor 8000
done
