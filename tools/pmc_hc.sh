# SQ counter pass over the HC kernel (developer aid); results under gpurun_out/pmc_hc*
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/gpurun_out/pmc_hc -o sq -- python $R/tools/prof_hc.py 1024 262144 60 9 > $R/gpurun_out/pmc_hc.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmc_hc2 -o sq2 -- python $R/tools/prof_hc.py 1024 262144 60 9 > $R/gpurun_out/pmc_hc2.log 2>&1
cd $R; for db in gpurun_out/pmc_hc/sq_results.db gpurun_out/pmc_hc2/sq2_results.db; do python tools/rocprof_summary.py $db | grep -A40 "PMC" | grep hc; done; tail -3 gpurun_out/pmc_hc2.log
