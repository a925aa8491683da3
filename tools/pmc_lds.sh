#!/bin/bash
# developer aid: LDS counters of the step's kernels (one rocprofv3 pass):  [LZ4AMD_LIB=<absolute path>] tools/pmc_lds.sh [tag]
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-lds}
B="python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-hc --no-extras --no-foreign"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/${tag}_pmc -o l -- $B > $R/gpurun_out/${tag}_pmc.log 2>&1
cd $R; python tools/rocprof_summary.py $(find gpurun_out/${tag}_pmc -name "*results.db" | head -1) | grep -E "k_decompress\(" | tee gpurun_out/${tag}_pmc.txt; find gpurun_out/${tag}_pmc -name "*.db" -delete
