python -m pytest tests/test_gpu_nets.py tests/test_gpu_ragged.py -m gpu -x -q -k "V2" 2>&1 | tail -3
python tools/gpu_profile_net.py V2_ch_rec 32 32 768 --ragged --wmin 520 --top 4 2>&1 | grep -v amdgpu.ids | head -8 | cut -c1-150
for W in 320 512 768 1024; do
  echo "== VSE_RAGGED_SELW=$W"
  for shape in "16 48 512 --wmin 330" "32 48 768 --wmin 520" "32 48 1280 --wmin 780"; do
    VSE_RAGGED_SELW=$W python tools/gpu_profile_net.py V4_ch_rec $shape --ragged --top 0 2>&1 | grep "ops, total"
  done
done
for W in 320 768; do for i in 1 2; do echo -n "bench SELW=$W: "; VSE_RAGGED_SELW=$W python bench.py --no-cpu-baseline --no-roofline --other-mode-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
