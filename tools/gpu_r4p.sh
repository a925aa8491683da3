#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for p in 60 20 90; do timeout 200 python tools/exp/accel_speed.py $p 2>&1 | grep acceleration; done
for shape in "256 4194304 60" "256 4194304 90" "256 4194304 20"; do
  ( LZ4AMD_LIB=variants/liblz4_amd_p1.so NOPROF=1 timeout 120 python tools/prof_dec.py $shape ) 2>&1 | grep -E "^decoder|Error|error" | sed "s/^/[p1] /"
done
