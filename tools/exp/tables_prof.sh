#!/bin/bash
# developer aid: linked 4 MiB blocks side by side, second copies from tables or not: per-dispatch kernel times
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD
for tb in 1 0; do
  export LZ4AMD_CHAIN_TABLES=$tb
  echo "== tables $tb"
  LZ4AMD_CHAIN_DEBUG=1 timeout 200 python tools/exp/linked_speed.py 1024 4096 60 2>&1 | grep -E "side by|lz4amd:"
  ( cd /tmp && rm -rf /tmp/tq$tb && timeout 300 rocprofv3 --kernel-trace -d /tmp/tq$tb -o x -- python $R/tools/exp/linked_speed.py 1024 4096 60 > /dev/null 2>&1 )
  db=$(find /tmp/tq$tb -name "*results.db" | head -1); python tools/exp/dispatch_times.py $db lz4amd | tail -8
done
