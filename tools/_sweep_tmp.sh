python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for hw in "1080 1920" "720 1280"; do python tools/parity_sweep.py 128 $hw 2>&1 | grep -v amdgpu.ids | tail -1; done
python tools/fuzz_graph.py --cases 200 --seed 41 --hilo --gpu 2>&1 | tail -1
python tools/fuzz_graph.py --cases 150 --seed 42 --gpu 2>&1 | tail -1
python tools/fuzz_ragged.py 2>&1 | tail -1
for i in 1 2 3; do python tools/dw_tile_check.py 2>&1 | grep -E "V4_ch_rec 56" | cut -c40-130; done
