#!/bin/bash
# one GPU visit: HC kernel time for variant builds of its tuning constants:  tools/gpu_hc_tune.sh [levels] variant...
lv=${1:-9}; shift
for v in "" "$@"; do
  if [ -n "$v" ]; then export LZ4AMD_LIB=variants/liblz4_amd_$v.so; else unset LZ4AMD_LIB; fi
  echo "== ${v:-product}"; timeout 120 python tools/prof_hc.py 4096 262144 60 $lv 2>&1 | grep -v "amdgpu.ids\|optimal"
done
