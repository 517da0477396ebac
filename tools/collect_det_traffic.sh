#!/bin/bash
# Runs on the GPU box: FETCH_SIZE / WRITE_SIZE passes over the mobile detector alone (tools/det_traffic.py), default and layer by layer.
# usage: bash tools/collect_det_traffic.sh r05   ->  gpurun_out/<tag>_fast_detector_traffic.json
R=$GRAFT_REPO_ROOT; TAG=${1:-r05}; M=${2:-V4_ch_det_fast}
cd /tmp; export TMPDIR=/tmp
for V in default layerwise; do
  F=""; [ $V = layerwise ] && F="--layerwise"
  for C in FETCH_SIZE WRITE_SIZE; do
    D=$R/gpurun_out/dt_${TAG}_${V}_$C; rm -rf $D
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o t -- python $R/tools/det_traffic.py run $M $F > /dev/null 2> $D.err
    rm -f $D/*kernel_trace.csv $D/*/*kernel_trace.csv
  done
done
cd $R
python - <<PY > gpurun_out/${TAG}_fast_detector_traffic.json
import json, subprocess, sys
out = {"model": "$M", "method": "tools/det_traffic.py: FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes over 4 forward passes of 64 x 544 x 960, "
       "detector alone; bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (units confirmed for these kernels' access shapes by profiles/r05_counter_calibration.json)"}
for v in ("default", "layerwise"):
    out[v] = json.loads(subprocess.run([sys.executable, "tools/det_traffic.py", "sum", f"gpurun_out/dt_${TAG}_{v}_FETCH_SIZE", f"gpurun_out/dt_${TAG}_{v}_WRITE_SIZE"],
                                       capture_output=True, text=True).stdout)
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/dt_${TAG}_*
head -12 gpurun_out/${TAG}_fast_detector_traffic.json
