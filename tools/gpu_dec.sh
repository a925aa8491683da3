#!/bin/bash
# decoder timing across shapes (+ optional A/B against the round-1 kernel) and the GPU tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
tag=${1:-dec}
for shape in "256 4194304 60" "2048 4194304 60" "4096 262144 60 9" "16384 65536 60"; do
  ( timeout 120 python tools/prof_dec.py $shape ) > gpurun_out/${tag}_v2.log 2>&1; grep -E "decoder|block total|pre-parse|copy wave" gpurun_out/${tag}_v2.log
done
if [ "${2:-}" = "tests" ]; then timeout 300 python -m pytest tests -m gpu -x -q --timeout 60 > gpurun_out/${tag}_tests.log 2>&1; tail -3 gpurun_out/${tag}_tests.log; fi
