#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_hints.py -m gpu -x -q --timeout 120 2>&1 | tail -2
for v in "" p2 p4; do
  lib=""; [ -n "$v" ] && lib="variants/liblz4_amd_$v.so"
  for shape in "256 4194304 60" "256 4194304 90" "256 4194304 20" "256 4194304 0" "4096 262144 60" "16384 65536 60"; do
    ( LZ4AMD_LIB=$lib NOPROF=1 timeout 120 python tools/prof_dec.py $shape ) 2>&1 | grep -E "^decoder|Error|error" | sed "s/^/[${v:-p3}] /"
  done
done
for v in pp; do for p in 60 90 20; do LZ4AMD_LIB=variants/liblz4_amd_$v.so timeout 120 python tools/prof_parser.py 256 4194304 $p 2>&1 | grep -v amdgpu.ids | sed "s/^/[$v] /"; done; done
( timeout 120 python tools/prof_dec.py 256 4194304 60 ) 2>&1 | tail -6
