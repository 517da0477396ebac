for ms in 3 4 6; do VSE_CHAIN_MAXSTAGES=$ms python tools/chain_check.py --time-only > gpurun_out/r4_c16_ms$ms.log 2>&1; done
VSE_CHAIN_MAXSTAGES=6 bash tools/trace_chain.sh V4_ch_det_fast 64 4 > gpurun_out/r4_c16_trace.log 2>&1
