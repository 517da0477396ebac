R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
run() { rm -rf $R/gpurun_out/pmc_$1; rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$1 -o p -- python $R/tools/gpu_profile_net.py V4_ch_det 16 544 960 --top 1 > /dev/null 2> $R/gpurun_out/pmc_$1.err; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU"
