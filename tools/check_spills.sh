#!/bin/bash
# Register spills / scratch of every kernel of the PRODUCT build (hipcc -Rpass-analysis=kernel-resource-usage): prints the kernels whose
# scratch size or spill count is not zero and exits 1 if there is one.  usage: bash tools/check_spills.sh [-DVSE_DEV_BUILD]
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/video-subtitle-extractor_amd/csrc; T=$(mktemp -d)
for f in $(python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; print(' '.join(g.HIP_SOURCES))"); do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -Rpass-analysis=kernel-resource-usage -c $C/$f -o $T/$f.o 2> $T/$f.log ) &
done
wait
python3 - "$T" <<'PY'
import glob, re, sys
bad = 0
tot = 0
for log in sorted(glob.glob(sys.argv[1] + "/*.log")):
    name = None
    cur = {}
    for ln in open(log):
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            name, cur = m.group(1), {}
            tot += 1
        for key in ("VGPRs Spill", "SGPRs Spill", "ScratchSize \[bytes/lane\]", "VGPRs", "Occupancy \[waves/SIMD\]"):
            m = re.search(key + r": (\d+)", ln)
            if m and name:
                cur[key] = int(m.group(1))
                if key.startswith("Occupancy"):
                    if cur.get("VGPRs Spill", 0) or cur.get("ScratchSize \\[bytes/lane\\]", 0):
                        print(f"SPILL {name}: {cur}")
                        bad += 1
print(f"{tot} kernels, {bad} with spills / scratch")
sys.exit(1 if bad else 0)
PY
rc=$?; rm -rf $T; exit $rc
