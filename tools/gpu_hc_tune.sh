#!/bin/bash
# one GPU visit: HC kernel time for variant builds of its tuning constants
for v in "" hc_refill8 hc_refill32 hc_batch2 hc_batch3 hc_run2 hc_run8; do
  if [ -n "$v" ]; then export LZ4AMD_LIB=variants/liblz4_amd_$v.so; else unset LZ4AMD_LIB; fi
  echo "== ${v:-product}"; timeout 120 python tools/prof_hc.py 4096 262144 60 9 2>&1 | grep "HC level"
done
