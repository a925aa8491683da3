#!/bin/bash
# HBM counters of the HC kernel alone (configs[3]), two passes:  tools/pmc_hc_only.sh TAG
cd "${GRAFT_REPO_ROOT:-/root/repo}"; tag=${1:-hc}; export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out
( cd /tmp && timeout 100 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_fetch_hc -o ${tag}hf -- python $R/tools/prof_hc.py 4096 262144 60 9 > /dev/null 2>&1 )
( cd /tmp && timeout 100 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_write_hc -o ${tag}hw -- python $R/tools/prof_hc.py 4096 262144 60 9 > /dev/null 2>&1 )
for db in $(find gpurun_out -name "${tag}*results.db"); do python tools/rocprof_summary.py $db > ${db%.db}.txt 2>&1; grep compress_hc ${db%.db}.txt; done
