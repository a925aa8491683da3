#!/bin/bash
# developer aid: instruction-cache counters of the step's kernels (one rocprofv3 pass):  tools/pmc_icache.sh [tag]
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-ic}
B="python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-hc --no-extras --no-foreign"
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/${tag}_pmc -o ic -- $B > $R/gpurun_out/${tag}_pmc.log 2>&1
cd $R; python tools/rocprof_summary.py $(find gpurun_out/${tag}_pmc -name "*results.db" | head -1) | grep -E "k_compress\(|k_decompress\(" | tee gpurun_out/${tag}_pmc.txt; find gpurun_out/${tag}_pmc -name "*.db" -delete
