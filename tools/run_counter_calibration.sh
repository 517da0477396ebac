#!/bin/bash
# Runs on the GPU box (via gpurun): the counter calibration microbenchmark under two separate --pmc passes.
# usage: bash tools/run_counter_calibration.sh r05     (then, back in the build container: python tools/counter_calibration.py r05)
R=$GRAFT_REPO_ROOT; TAG=${1:-r05}
D=$R/gpurun_out/calib_$TAG
rm -rf $D; mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $D/counter_calib $R/tools/ubench/counter_calib.hip || exit 1
$D/counter_calib > $D/known.json || exit 1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch -o calib -- $D/counter_calib > /dev/null 2> $D/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/write -o calib -- $D/counter_calib > /dev/null 2> $D/write.err
rm -f $D/*/calib_kernel_trace.csv $D/*/*agent_info.csv $D/counter_calib
ls -la $D $D/fetch $D/write
head -3 $D/fetch/calib_counter_collection.csv
