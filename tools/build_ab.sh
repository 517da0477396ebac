#!/bin/bash
# Development-build variants of ONE source for same-box A/B runs: every csrc/*.hip compiled once with -DVSE_DEV_BUILD into build/devobj,
# then one library per variant with that source recompiled under the variant's -D flags:
#   bash tools/build_ab.sh conv_c3.hip C3ABL0="-DVSE_C3_ABL=0" C3ABL1="-DVSE_C3_ABL=1" ...   ->  build/ab/libvse_<NAME>.so   (use with VSE_LIB_PATH)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/video-subtitle-extractor_amd/csrc
SRC=$1; shift
mkdir -p $R/build/devobj $R/build/ab
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DVSE_DEV_BUILD"
for s in $C/*.hip; do
  o=$R/build/devobj/$(basename $s).o
  if [ ! -f $o ] || [ $s -nt $o ] || [ $C/common.h -nt $o ] || [ $C/conv_common.h -nt $o ] || [ $R/include/vse_hip.h -nt $o ]; then
    /opt/rocm/bin/hipcc $FL -c $s -o $o &
  fi
done
wait
for kv in "$@"; do
  name=${kv%%=*}; flags=${kv#*=}
  /opt/rocm/bin/hipcc $FL $flags -c $C/$SRC -o $R/build/ab/${SRC%.hip}_$name.o &
done
wait
for kv in "$@"; do
  name=${kv%%=*}
  objs=$(ls $R/build/devobj/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/ab/libvse_$name.so $objs $R/build/ab/${SRC%.hip}_$name.o
  echo built build/ab/libvse_$name.so
done
