python tools/chain_check.py > gpurun_out/r4_c08_check.log 2>&1; tail -10 gpurun_out/r4_c08_check.log
for ms in 2 3 6; do VSE_CHAIN_MAXSTAGES=$ms python tools/chain_check.py --time-only > gpurun_out/r4_c08_ms$ms.log 2>&1; done
VSE_CHAIN_MAXSTAGES=6 bash tools/trace_chain.sh V4_ch_det_fast 64 6 > gpurun_out/r4_c08_trace.log 2>&1
