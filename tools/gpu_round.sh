#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats, HBM counters (separate passes).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
tag=${1:-r01}
timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gputests.log 2>&1; tail -2 gpurun_out/${tag}_gputests.log
timeout 400 python bench.py --steps 10 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 600 gpurun_out/${tag}_bench.json
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof -o ${tag} -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-hc > $R/gpurun_out/${tag}_prof.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_fetch -o ${tag}f -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-hc > $R/gpurun_out/${tag}_pmc_fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_write -o ${tag}w -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-hc > $R/gpurun_out/${tag}_pmc_write.log 2>&1 )
# the HC side measurement (configs[3]) profiled on its own, so that the per-kernel averages above are those of the step
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_hc -o ${tag}hc -- python $R/tools/prof_hc.py 4096 262144 60 9 > $R/gpurun_out/${tag}_prof_hc.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_fetch_hc -o ${tag}hf -- python $R/tools/prof_hc.py 1024 262144 60 9 > /dev/null 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_write_hc -o ${tag}hw -- python $R/tools/prof_hc.py 1024 262144 60 9 > /dev/null 2>&1 )
find gpurun_out -name "*.db" | head
for db in $(find gpurun_out -name "${tag}*results.db"); do python tools/rocprof_summary.py $db > ${db%.db}.txt 2>&1; tail -n 12 ${db%.db}.txt; done
du -sh gpurun_out
