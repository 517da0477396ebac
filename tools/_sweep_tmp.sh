python tools/chain_check.py --time > gpurun_out/r4_c22_check.log 2>&1; grep -E "chain_check|64x544x960" gpurun_out/r4_c22_check.log | cut -c1-120
python -m pytest tests/test_gpu_chain.py tests/test_gpu_nets.py tests/test_gpu_pipeline.py tests/test_gpu_large.py -x -q 2>&1 | tail -3
for hw in "1080 1920" "720 1280"; do echo "== default $hw"; python tools/parity_sweep.py 128 $hw 2>&1 | grep -v amdgpu.ids | tail -3; done > gpurun_out/r4_c22_parity.log 2>&1
cat gpurun_out/r4_c22_parity.log
python tools/gpu_profile_net.py V4_ch_det_fast 64 544 960 --hilo --top 100 > gpurun_out/r4_c22_prof_V4_fast_chained.log 2>&1
VSE_CHAIN=0 python tools/gpu_profile_net.py V4_ch_det_fast 64 544 960 --hilo --top 100 > gpurun_out/r4_c22_prof_V4_fast_layerwise.log 2>&1
