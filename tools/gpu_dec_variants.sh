#!/bin/bash
# developer aid: decoder kernel time of the product library and of variants (with tables P60 / P90 / HC level 9 blocks / without tables):  tools/gpu_dec_variants.sh name1 ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in product "$@"; do
  [ "$v" = product ] && unset LZ4AMD_LIB || export LZ4AMD_LIB=variants/liblz4_amd_$v.so
  echo "== $v"
  NOPROF=1 timeout 60 python tools/prof_dec.py 256 4194304 60 0 2>&1 | grep "^decoder" | cut -c1-150
  NOPROF=1 timeout 60 python tools/prof_dec.py 256 4194304 90 0 2>&1 | grep "^decoder" | cut -c1-150
  NOPROF=1 timeout 60 python tools/prof_dec.py 4096 262144 60 9 2>&1 | grep "^decoder" | cut -c1-150
  NOPROF=1 NOHINTS=1 timeout 60 python tools/prof_dec.py 256 4194304 60 0 2>&1 | grep "^decoder" | cut -c1-150
  timeout 60 python tools/prof_refdec.py 60 2>&1 | grep "^reference"
done
