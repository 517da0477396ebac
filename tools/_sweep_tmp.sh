python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -s -k test_box_parity_rate_real_detector 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r4_c27_test.log; cat gpurun_out/r4_c27_test.log
for i in 1 2; do python bench.py --no-cpu-baseline > gpurun_out/r4_c27_bench$i.json 2> gpurun_out/r4_c27_bench$i.err; python -c "
import json; d=json.load(open('gpurun_out/r4_c27_bench$i.json')); print(d['value'], d['ms_per_step'], {k:v.get('value') for k,v in d['config']['secondary'].items()})"; done
