bash tools/collect_profiles.sh r04 > gpurun_out/r4_c44_prof.log 2>&1
bash tools/collect_profiles.sh r04_fast --models fast > gpurun_out/r4_c44_prof_fast.log 2>&1
tail -3 gpurun_out/r4_c44_prof.log | cut -c1-300; tail -3 gpurun_out/r4_c44_prof_fast.log | cut -c1-300
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for m in V4_ch_det_fast; do for v in "" "--layerwise" "--plain"; do t=$(echo "$m$v" | tr -d ' -'); 
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/tf_$t -o t -- python $R/tools/det_traffic.py run $m $v > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/tw_$t -o t -- python $R/tools/det_traffic.py run $m $v > /dev/null 2>&1
python $R/tools/det_traffic.py sum /tmp/tf_$t /tmp/tw_$t > $R/gpurun_out/r4_c44_traffic_$t.json; head -8 $R/gpurun_out/r4_c44_traffic_$t.json | tail -4; done; done
cd $R
python tools/gpu_profile_net.py V4_ch_det_fast 64 544 960 --hilo --no-chain --top 76 2>&1 | grep -v amdgpu.ids | cut -c1-175 > gpurun_out/r4_c44_prof_V4_layerwise.log
python tools/gpu_profile_net.py V4_ch_det_fast 64 544 960 --hilo --top 76 2>&1 | grep -v amdgpu.ids | cut -c1-175 > gpurun_out/r4_c44_prof_V4_default.log
python tools/gpu_profile_net.py V3_ch_det_fast 64 544 960 --hilo --no-chain --top 90 2>&1 | grep -v amdgpu.ids | cut -c1-175 > gpurun_out/r4_c44_prof_V3_layerwise.log
python tools/gpu_profile_net.py V3_ch_det_fast 64 544 960 --hilo --top 90 2>&1 | grep -v amdgpu.ids | cut -c1-175 > gpurun_out/r4_c44_prof_V3_default.log
python tools/gpu_profile_net.py V4_ch_rec_fast 56 48 896 --ragged --top 60 2>&1 | grep -v amdgpu.ids | cut -c1-175 > gpurun_out/r4_c44_prof_rec_fast.log
python tools/gpu_profile_net.py V4_ch_det 64 544 960 --top 100 2>&1 | grep -v amdgpu.ids | cut -c1-175 > gpurun_out/r4_c44_prof_det.log
python tools/gpu_profile_net.py V4_ch_rec 56 48 896 --ragged --top 80 2>&1 | grep -v amdgpu.ids | cut -c1-175 > gpurun_out/r4_c44_prof_rec.log
