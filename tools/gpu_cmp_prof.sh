#!/bin/bash
# one GPU visit: compressor phase stamps, stride 2 / stride 4, P60 / P20
for a in 1 2; do for p in 60 20; do
  timeout 300 python tools/prof_cmp.py 256 $p $a
  LZ4AMD_LIB=variants/liblz4_amd_match.so timeout 300 python tools/prof_cmp.py 256 $p $a
done; done 2>&1 | grep -v amdgpu.ids
