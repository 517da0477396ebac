python tools/chain_check.py --time > gpurun_out/r4_c43_check.log 2>&1; grep -E "chain_check|FAIL|64x544x960" gpurun_out/r4_c43_check.log | cut -c1-120
python tools/dw_tile_check.py 2>&1 | grep -v amdgpu.ids | cut -c1-150
python bench.py --no-cpu-baseline > gpurun_out/r4_c43_bench.json 2> gpurun_out/r4_c43_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r4_c43_bench.json')); print(d['value'], d['ms_per_step'], {k:v.get('value') for k,v in d['config']['secondary'].items()})"
grep "per-net" gpurun_out/r4_c43_bench.err
