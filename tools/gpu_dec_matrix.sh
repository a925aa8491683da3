#!/bin/bash
# developer aid: decoder kernel time of variants over the shapes that matter:  tools/gpu_dec_matrix.sh name1 ...   (product = the library itself)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in "$@"; do
  [ "$v" = product ] && unset LZ4AMD_LIB || export LZ4AMD_LIB=variants/liblz4_amd_$v.so
  echo "== $v"
  for a in "256 4194304 20 0" "256 4194304 60 0" "256 4194304 90 0" "4096 262144 60 9" "16384 65536 60 0"; do NOPROF=1 timeout 60 python tools/prof_dec.py $a 2>&1 | grep "^decoder" | cut -c1-110; done
  NOPROF=1 NOHINTS=1 timeout 60 python tools/prof_dec.py 256 4194304 60 0 2>&1 | grep "^decoder" | cut -c1-110
  NOPROF=1 NOHINTS=1 timeout 60 python tools/prof_dec.py 16384 65536 60 0 2>&1 | grep "^decoder" | cut -c1-110
  timeout 60 python tools/prof_refdec.py 60 2>&1 | grep "^reference"
done
