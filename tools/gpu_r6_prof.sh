#!/bin/bash
# round 6 developer aid: the decoder's role profile (LZ4AMD_PROF stamps) for the product and variants
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
tag=${1:-r06prof}; shift
for v in product "$@"; do
  [ "$v" = product ] && unset LZ4AMD_LIB || export LZ4AMD_LIB=variants/liblz4_amd_$v.so
  echo "== $v"
  for args in "256 4194304 60 0" "256 4194304 90 0" "4096 262144 60 9"; do
    timeout 60 python tools/prof_dec.py $args 2>&1 | cut -c1-200
  done
done > gpurun_out/${tag}.log 2>&1
cat gpurun_out/${tag}.log
