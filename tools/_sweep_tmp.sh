python tools/chain_check.py --time > gpurun_out/r4_c26_check.log 2>&1; grep -E "chain_check|FAIL|64x544x960" gpurun_out/r4_c26_check.log | cut -c1-120
for hw in "1080 1920" "720 1280"; do echo "== default $hw"; python tools/parity_sweep.py 128 $hw 2>&1 | grep -v amdgpu.ids | tail -3; done > gpurun_out/r4_c26_parity.log 2>&1
cat gpurun_out/r4_c26_parity.log
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r4_c26_tests.log; cat gpurun_out/r4_c26_tests.log
python tools/fuzz_graph.py --cases 200 --seed 9 --hilo --gpu 2>&1 | tail -2
python bench.py > gpurun_out/r4_c26_bench.json 2> gpurun_out/r4_c26_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r4_c26_bench.json')); print(d['value'], d['ms_per_step'], {k:v.get('value') for k,v in d['config']['secondary'].items()})"
