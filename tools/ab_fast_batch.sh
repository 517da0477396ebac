run() { echo -n "$*: "; python bench.py "$@" --no-cpu-baseline --no-roofline --other-mode-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run --models fast-real --steps 40
run --models fast-real --steps 20 --batch 128
run --models fast-real --steps 80 --batch 32
run --models fast-real --steps 40 --rec-streams 1
run --models fast-real --steps 40 --rec-streams 3
