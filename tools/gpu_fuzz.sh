#!/bin/bash
# developer aid: the reference's own fuzzers (linked against the library: oracle/_ref/*_amd) with several seeds, longer than the test suite runs them
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export LD_LIBRARY_PATH=$PWD/lz4_amd:$LD_LIBRARY_PATH
for s in ${SEEDS:-11 222 3333 44444}; do
  timeout 120 oracle/_ref/frametest_amd -s$s -T${T:-12}s 2>&1 | tail -2 | tr '\n' ' '; echo " [frametest seed $s rc=$?]"
  timeout 120 oracle/_ref/fuzzer_amd -s$s -T${T:-12}s 2>&1 | tail -2 | tr '\n' ' '; echo " [fuzzer seed $s rc=$?]"
done
