#!/bin/bash
# conv_gemm_kernel tile configurations forced one at a time (development build: VSE_GEMM_CFG) on the server detector: per-op times of the
# 1x1 layers.  usage (GPU box): bash tools/ab_gemm_cfg.sh path/to/libvse_hip_dev.so
L=${1:-build/devlib/libvse_hip_dev.so}
for cfg in "" 0 6 16 17 18 19; do
  echo "== VSE_GEMM_CFG=$cfg"
  env VSE_DEV_BUILD=1 VSE_LIB_PATH=$L VSE_GEMM_CFG=$cfg python tools/gpu_profile_net.py V4_ch_det 64 544 960 --top 100 2>&1 | grep "N=64 544x960\|k1x1 s1 cin\(896\|1216\|256\|512\|768\|1664\|1920\|2112\|1024\) " | cut -c1-60,118-215
done
