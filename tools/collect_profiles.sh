#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel stats + separate PMC passes, all under gpurun_out/.
R=$GRAFT_REPO_ROOT; TAG=${1:-r01}; shift; EXTRA="$@"      # EXTRA: more bench.py flags (e.g. --models fast for the mobile pair: tag r04_fast)
cd $R && python bench.py $EXTRA > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG $R/gpurun_out/pmc_fetch_$TAG $R/gpurun_out/pmc_write_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py $EXTRA --no-cpu-baseline --no-secondary --other-mode-steps 0 > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_$TAG.err
# the same bench with one batch at a time and one recogniser stream: kernel durations without overlap (what roofline.avg_launch_us measures)
rm -rf $R/gpurun_out/prof_${TAG}_seq
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_seq -o bench -- python $R/bench.py $EXTRA --no-overlap --rec-streams 1 --no-cpu-baseline --no-secondary --other-mode-steps 0 > $R/gpurun_out/prof_${TAG}_seq_bench.json 2> $R/gpurun_out/prof_${TAG}_seq.err
rm -f $R/gpurun_out/prof_${TAG}_seq/bench_kernel_trace.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -o bench -- python $R/bench.py $EXTRA --steps 1 --warmup 1 --no-overlap --rec-streams 1 --no-cpu-baseline --no-roofline --other-mode-steps 0 > /dev/null 2> $R/gpurun_out/pmc_fetch_$TAG.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_$TAG -o bench -- python $R/bench.py $EXTRA --steps 1 --warmup 1 --no-overlap --rec-streams 1 --no-cpu-baseline --no-roofline --other-mode-steps 0 > /dev/null 2> $R/gpurun_out/pmc_write_$TAG.err
rm -rf $R/gpurun_out/pmc_sq_$TAG
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_$TAG -o bench -- python $R/bench.py $EXTRA --steps 1 --warmup 1 --no-overlap --rec-streams 1 --no-cpu-baseline --no-roofline --other-mode-steps 0 > /dev/null 2> $R/gpurun_out/pmc_sq_$TAG.err
rm -f $R/gpurun_out/prof_$TAG/bench_kernel_trace.csv   # large; the stats CSV is what gets committed
rm -f $R/gpurun_out/pmc_*_$TAG/bench_kernel_trace.csv $R/gpurun_out/pmc_*_$TAG/*agent_info.csv
du -sh $R/gpurun_out
cd $R; cat gpurun_out/bench_$TAG.json; head -5 gpurun_out/prof_$TAG/bench_kernel_stats.csv | cut -c1-160
