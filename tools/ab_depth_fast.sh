run() { echo -n "$*: "; python bench.py "$@" --no-cpu-baseline --no-roofline --other-mode-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run --models fast-real --steps 40
run --models fast-real --steps 40 --det-depth 3
run --models fast-real --steps 40 --det-depth 4
run --models fast-real --steps 40 --det-depth 3 --rec-span 3
run --models fast-real --steps 40 --det-depth 4 --rec-span 4
done
