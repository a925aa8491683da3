#!/bin/bash
# round 4: decoder with entry-point tables - parser profile and timing by shape
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
tag=${1:-r4b}
timeout 300 python -m pytest tests/test_gpu_hints.py -m gpu -x -q --timeout 120 > gpurun_out/${tag}_hinttests.log 2>&1; tail -3 gpurun_out/${tag}_hinttests.log
for shape in "256 4194304 60" "256 4194304 20" "256 4194304 90" "256 4194304 0" "4096 262144 60" "16384 65536 60"; do
  ( NOPROF=1 timeout 120 python tools/prof_dec.py $shape ) 2>&1 | grep -E "^decoder|Error|error"
done
for p in 60 90 20; do ( timeout 120 python tools/prof_dec.py 256 4194304 $p ) > gpurun_out/${tag}_profdec$p.log 2>&1; tail -7 gpurun_out/${tag}_profdec$p.log; done
