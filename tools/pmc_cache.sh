# L2 / texture-path counters per kernel for the detector (runs on the GPU box via gpurun); outputs under gpurun_out/pmc_{c,d}
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
run() { rm -rf $R/gpurun_out/pmc_$1; rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$1 -o p -- python $R/tools/gpu_profile_net.py V4_ch_det 64 544 960 --top 1 > /dev/null 2> $R/gpurun_out/pmc_$1.err; tail -2 $R/gpurun_out/pmc_$1.err | cut -c1-200; }
run c "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
run d "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum"
