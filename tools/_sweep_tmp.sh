python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r4_c34_tests.log; cat gpurun_out/r4_c34_tests.log
python -m pytest tests/test_gpu_bench.py -m gpu -q -s -k c2_against 2>&1 | grep -E "C2 path|passed|failed|Error|assert" | head
