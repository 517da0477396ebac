python tools/chain_check.py --time > gpurun_out/r4_c32_check.log 2>&1; grep -E "chain_check|FAIL|64x544x960" gpurun_out/r4_c32_check.log | cut -c1-120
python tools/gpu_profile_net.py V4_ch_det_fast 64 544 960 --hilo --top 40 2>&1 | grep -v amdgpu.ids | cut -c1-175 > gpurun_out/r4_c32_prof_V4_default.log
python bench.py --no-cpu-baseline > gpurun_out/r4_c32_bench.json 2> gpurun_out/r4_c32_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r4_c32_bench.json')); print(d['value'], d['ms_per_step'], {k:v.get('value') for k,v in d['config']['secondary'].items()})"
grep "per-net" gpurun_out/r4_c32_bench.err
