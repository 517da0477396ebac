python tools/chain_check.py --time-only 2>&1 | grep "64x544x960" | cut -c1-120
python bench.py --no-cpu-baseline > gpurun_out/r4_c39_bench.json 2> gpurun_out/r4_c39_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r4_c39_bench.json')); print(d['value'], d['ms_per_step'], {k:v.get('value') for k,v in d['config']['secondary'].items()})"
grep "per-net" gpurun_out/r4_c39_bench.err
python tools/gpu_profile_net.py V4_ch_det_fast 64 544 960 --hilo --no-chain --top 76 2>&1 | grep -v amdgpu.ids | cut -c1-175 > gpurun_out/r4_c39_prof_V4_layerwise.log
