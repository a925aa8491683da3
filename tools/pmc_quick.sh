#!/bin/bash
# developer aid: SQ instruction counters of the step's two kernels (one rocprofv3 pass), printed:  tools/pmc_quick.sh [tag] [extra bench flags]
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-q}; shift
B="python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-hc --no-extras --no-foreign $*"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $R/gpurun_out/${tag}_pmc -o s1 -- $B > $R/gpurun_out/${tag}_pmc.log 2>&1
cd $R; python tools/rocprof_summary.py $(find gpurun_out/${tag}_pmc -name "*results.db" | head -1) | grep -E "k_compress\(|k_decompress\(" | tee gpurun_out/${tag}_pmc.txt
