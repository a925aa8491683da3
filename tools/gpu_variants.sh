#!/bin/bash
# developer aid: compress kernel time of the product library and of every variants/liblz4_amd_<name>.so given (P60 unless P= is set):  tools/gpu_variants.sh name1 name2 ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
P=${P:-60}
echo "== product"; timeout 100 python tools/prof_cmp.py 256 $P 1 2>&1 | grep -v amdgpu.ids | head -${LINES_EACH:-1}
for v in "$@"; do echo "== $v"; LZ4AMD_LIB=variants/liblz4_amd_$v.so timeout 100 python tools/prof_cmp.py 256 $P 1 2>&1 | grep -v amdgpu.ids | head -${LINES_EACH:-1}; done
