#!/bin/bash
# SQ counters of the compress / decompress kernels (two passes), summary into gpurun_out/<tag>_sq.txt
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-sq}
B="python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-hc --no-extras"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $R/gpurun_out/${tag}_pmc1 -o s1 -- $B > $R/gpurun_out/${tag}_pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/${tag}_pmc2 -o s2 -- $B > $R/gpurun_out/${tag}_pmc2.log 2>&1
cd $R; for db in $(find gpurun_out/${tag}_pmc1 gpurun_out/${tag}_pmc2 -name "*results.db"); do python tools/rocprof_summary.py $db | grep -A40 "PMC"; done > gpurun_out/${tag}_sq.txt 2>&1; cat gpurun_out/${tag}_sq.txt | grep -E "k_compress |k_decompress" 
