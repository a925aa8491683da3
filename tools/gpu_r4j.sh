#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
tag=${1:-r4j}
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/${tag}_gputests.log 2>&1; tail -3 gpurun_out/${tag}_gputests.log
for shape in "256 4194304 60" "256 4194304 90" "256 4194304 20" "256 4194304 0" "4096 262144 60" "16384 65536 60"; do
  ( NOPROF=1 timeout 120 python tools/prof_dec.py $shape ) 2>&1 | grep -E "^decoder|Error|error"
done
for p in 60 90 20; do LZ4AMD_LIB=variants/liblz4_amd_pp.so timeout 120 python tools/prof_parser.py 256 4194304 $p 2>&1 | grep -v amdgpu.ids; done
( timeout 120 python tools/prof_dec.py 256 4194304 60 ) 2>&1 | tail -6
( NOPROF=1 timeout 120 python tools/prof_cmp.py 256 4194304 60 ) 2>&1 | tail -3
