export VSE_CHAIN_HEAD=0
python tools/chain_check.py --time 2>&1 | grep -E "chain_check|FAIL|64x544x960" | cut -c1-120
for hw in "1080 1920" "720 1280"; do python tools/parity_sweep.py 128 $hw 2>&1 | grep -v amdgpu.ids | tail -1; done
