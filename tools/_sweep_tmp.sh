python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r4_c49_tests.log; cat gpurun_out/r4_c49_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
