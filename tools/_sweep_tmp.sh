bash tools/collect_profiles.sh r04 > gpurun_out/r4_c52_prof.log 2>&1
tail -2 gpurun_out/r4_c52_prof.log | cut -c1-200
python -c "
import json; d=json.load(open('gpurun_out/bench_r04.json')); print(d['value'], d['ms_per_step'], {k:(v.get('value'), v.get('timed_blocks')) for k,v in d['config']['secondary'].items()})"
