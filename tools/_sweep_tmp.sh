python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r4_c45_tests.log; cat gpurun_out/r4_c45_tests.log
python tools/fuzz_graph.py --cases 250 --seed 21 --hilo --gpu 2>&1 | tail -1
python tools/fuzz_graph.py --cases 150 --seed 22 --gpu 2>&1 | tail -1
python tools/fuzz_ragged.py 2>&1 | tail -1
for hw in "1080 1920" "720 1280"; do python tools/parity_sweep.py 128 $hw 2>&1 | grep -v amdgpu.ids | tail -1; done
python bench.py > gpurun_out/r4_c45_bench.json 2> gpurun_out/r4_c45_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r4_c45_bench.json')); print(d['value'], d['ms_per_step'], {k:v.get('value') for k,v in d['config']['secondary'].items()}, d['roofline']['frac'], d['roofline']['detector_convs'], d['cpu_baseline']['value'])"
