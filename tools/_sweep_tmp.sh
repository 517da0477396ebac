python tools/chain_check.py --time > gpurun_out/r4_c18_check.log 2>&1; grep -E "chain_check|64x544x960" gpurun_out/r4_c18_check.log | cut -c1-120
python -m pytest tests/test_gpu_chain.py tests/test_gpu_nets.py -x -q 2>&1 | tail -3
for hw in "1080 1920" "720 1280"; do echo "== default $hw"; python tools/parity_sweep.py 128 $hw 2>&1 | grep -v amdgpu.ids | tail -3; done > gpurun_out/r4_c18_parity.log 2>&1
cat gpurun_out/r4_c18_parity.log
