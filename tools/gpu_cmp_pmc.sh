#!/bin/bash
# one GPU visit: SQ_INSTS_VALU / SALU / LDS of the compress kernel for variant builds that skip phases
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in "" noemit noselect noprobe; do
  if [ -n "$v" ]; then export LZ4AMD_LIB=$R/variants/liblz4_amd_$v.so; else unset LZ4AMD_LIB; fi
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/cx_$v -o cx -- python $R/tools/exp/cmp_run.py 60 2>&1 | grep "ms$"
  python $R/tools/rocprof_summary.py $R/gpurun_out/cx_$v/cx_results.db | grep "k_compress(" 
done
