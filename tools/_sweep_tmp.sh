python tools/chain_check.py --time > gpurun_out/r4_c24_check.log 2>&1; grep -E "chain_check|FAIL|64x544x960" gpurun_out/r4_c24_check.log | cut -c1-120
for hw in "1080 1920" "720 1280"; do echo "== default $hw"; python tools/parity_sweep.py 128 $hw 2>&1 | grep -v amdgpu.ids | tail -3; done > gpurun_out/r4_c24_parity.log 2>&1
cat gpurun_out/r4_c24_parity.log
for m in V4_ch_det_fast V3_ch_det_fast; do
python tools/gpu_profile_net.py $m 64 544 960 --hilo --top 100 > gpurun_out/r4_c24_prof_${m}_default.log 2>&1
VSE_CHAIN=0 VSE_CHAIN_LO=0 python tools/gpu_profile_net.py $m 64 544 960 --hilo --top 100 > gpurun_out/r4_c24_prof_${m}_layerwise.log 2>&1
done
grep -E "total|conv_dwpw" gpurun_out/r4_c24_prof_V4_ch_det_fast_layerwise.log | head -12
python -m pytest tests/test_gpu_chain.py tests/test_gpu_nets.py tests/test_gpu_pipeline.py tests/test_gpu_large.py -x -q 2>&1 | tail -3
