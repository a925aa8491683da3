#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats, HBM counters (separate passes), SQ counters.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
tag=${1:-r06}
what=${2:-all}
if [ "$what" != "prof" ]; then
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/${tag}_gputests.log 2>&1; tail -2 gpurun_out/${tag}_gputests.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench.json'))
print({k: d[k] for k in ('value','ms_per_step','compress_GBps','decompress_GBps','ratio')}); print(d['roofline']); print(d['roofline_decompress'])"
[ "$what" = "quick" ] && exit 0
fi
export TMPDIR=/tmp
R=$PWD
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-hc --no-extras --no-foreign"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof -o ${tag} -- $B > $R/gpurun_out/${tag}_prof.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_fetch -o ${tag}f -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-hc --no-extras --no-foreign > $R/gpurun_out/${tag}_pmc_fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_write -o ${tag}w -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-hc --no-extras --no-foreign > $R/gpurun_out/${tag}_pmc_write.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $R/gpurun_out/${tag}_pmc_sq1 -o ${tag}s1 -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-hc --no-extras --no-foreign > $R/gpurun_out/${tag}_pmc_sq1.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/${tag}_pmc_sq2 -o ${tag}s2 -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-hc --no-extras --no-foreign > $R/gpurun_out/${tag}_pmc_sq2.log 2>&1 )
# the decoder on plain LZ4 blocks: the step's own blocks without their tables (--no-hints), and blocks the reference compressed (tools/prof_refdec.py)
NH="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-hc --no-extras --no-foreign --no-hints"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_nohints -o ${tag}nh -- $NH > $R/gpurun_out/${tag}_prof_nohints.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_fetch_nohints -o ${tag}nhf -- $NH > /dev/null 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_write_nohints -o ${tag}nhw -- $NH > /dev/null 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $R/gpurun_out/${tag}_pmc_sq1_nohints -o ${tag}nhs -- $NH > /dev/null 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_refdec -o ${tag}rd -- python $R/tools/prof_refdec.py 60 > $R/gpurun_out/${tag}_prof_refdec.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_fetch_refdec -o ${tag}rdf -- python $R/tools/prof_refdec.py 60 > /dev/null 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_write_refdec -o ${tag}rdw -- python $R/tools/prof_refdec.py 60 > /dev/null 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $R/gpurun_out/${tag}_pmc_sq1_refdec -o ${tag}rds -- python $R/tools/prof_refdec.py 60 > /dev/null 2>&1 )
# the HC side measurement (configs[3]) profiled on its own, so that the per-kernel averages above are those of the step
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_hc -o ${tag}hc -- python $R/tools/prof_hc.py 4096 262144 60 9 > $R/gpurun_out/${tag}_prof_hc.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_fetch_hc -o ${tag}hf -- python $R/tools/prof_hc.py 4096 262144 60 9 > /dev/null 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_write_hc -o ${tag}hw -- python $R/tools/prof_hc.py 4096 262144 60 9 > /dev/null 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $R/gpurun_out/${tag}_pmc_sq1_hc -o ${tag}hs -- python $R/tools/prof_hc.py 4096 262144 60 9 > /dev/null 2>&1 )
# configs[2] device resident: the chained decode of linked blocks, the gather into frame layout, the batched XXH32 (so that every frac of the line has a profiles/ counterpart)
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_frame -o ${tag}fr -- python $R/tools/prof_frame.py > $R/gpurun_out/${tag}_prof_frame.log 2>&1 )
( timeout 120 python tools/stress_gpu.py 30 5 2>&1 | tail -1 ) > gpurun_out/${tag}_stress.log; cat gpurun_out/${tag}_stress.log
( timeout 200 python tools/stress_linked.py 60 5 2>&1 | tail -1 ) > gpurun_out/${tag}_stress_linked.log; cat gpurun_out/${tag}_stress_linked.log
for db in $(find gpurun_out -name "${tag}*results.db"); do python tools/rocprof_summary.py $db > ${db%.db}.txt 2>&1; tail -n 6 ${db%.db}.txt; rm -f $db; done      # (the databases are tens of MB: gpurun copies back 64 MiB at most)
find gpurun_out -type f \( -name "*.db" -o -name "*.pftrace" -o -name "*.rocpd" \) -delete
du -sh gpurun_out
