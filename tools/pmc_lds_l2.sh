#!/bin/bash
# LDS bank conflicts and L2 hit rate per kernel (one batch at a time): two more --pmc passes beside tools/collect_profiles.sh
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_lds $R/gpurun_out/pmc_l2
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_lds -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-overlap --rec-streams 1 --no-cpu-baseline --no-roofline > /dev/null 2> $R/gpurun_out/pmc_lds.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_l2 -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-overlap --rec-streams 1 --no-cpu-baseline --no-roofline > /dev/null 2> $R/gpurun_out/pmc_l2.err
rm -f $R/gpurun_out/pmc_lds/bench_kernel_trace.csv $R/gpurun_out/pmc_l2/bench_kernel_trace.csv
tail -2 $R/gpurun_out/pmc_lds.err $R/gpurun_out/pmc_l2.err
ls -la $R/gpurun_out/pmc_lds $R/gpurun_out/pmc_l2
