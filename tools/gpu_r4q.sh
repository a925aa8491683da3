#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for shape in "256 4194304 60" "256 4194304 90" "256 4194304 20" "4096 262144 60" "16384 65536 60"; do
  ( NOPROF=1 timeout 120 python tools/prof_dec.py $shape ) 2>&1 | grep -E "^decoder|Error|error"
done
for shape in "256 4194304 60" "256 4194304 90" "16384 65536 60"; do
( NOPROF=1 NOHINTS=1 timeout 120 python tools/prof_dec.py $shape ) 2>&1 | grep -E "^decoder|Error|error"
done
( timeout 120 python tools/prof_dec.py 256 4194304 60 ) 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hints.py -m gpu -x -q --timeout 300 2>&1 | tail -2
