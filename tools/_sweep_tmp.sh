python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r4_c19_tests.log; cat gpurun_out/r4_c19_tests.log
python bench.py > gpurun_out/r4_c19_bench.json 2> gpurun_out/r4_c19_bench.err; tail -c 2500 gpurun_out/r4_c19_bench.json; grep -E "secondary|per-net" gpurun_out/r4_c19_bench.err
