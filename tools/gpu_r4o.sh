#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_frame.py -m gpu -x -q --timeout 120 2>&1 | tail -2
for v in "" p2 p4; do
  lib=""; [ -n "$v" ] && lib="variants/liblz4_amd_$v.so"
  for shape in "256 4194304 60" "256 4194304 90" "256 4194304 20" "4096 262144 60" "16384 65536 60" "2048 4194304 60"; do
    ( LZ4AMD_LIB=$lib NOPROF=1 timeout 120 python tools/prof_dec.py $shape ) 2>&1 | grep -E "^decoder|Error|error" | sed "s/^/[${v:-p3}] /"
  done
done
