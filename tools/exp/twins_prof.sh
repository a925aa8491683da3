#!/bin/bash
# developer aid: linked blocks side by side with / without twins, per-dispatch kernel times
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD
for tw in 1 0; do
  export LZ4AMD_CHAIN_TWINS=$tw
  echo "== twins $tw"
  timeout 200 python tools/exp/linked_speed.py 1024 4096 60 2>&1 | grep "side by"
  timeout 200 python tools/exp/linked_speed.py 256 4096 60 2>&1 | grep "side by"
  timeout 200 python tools/exp/linked_speed.py 256 64 60 2>&1 | grep "side by"
  timeout 200 python tools/exp/linked_speed.py 256 256 60 2>&1 | grep "side by"
  ( cd /tmp && rm -rf /tmp/tp$tw && timeout 300 rocprofv3 --kernel-trace -d /tmp/tp$tw -o x -- python $R/tools/exp/linked_speed.py 1024 4096 60 > /dev/null 2>&1 )
  db=$(find /tmp/tp$tw -name "*results.db" | head -1); python tools/exp/dispatch_times.py $db lz4amd | tail -7
done
