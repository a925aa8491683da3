#!/bin/bash
# round 6 developer aid: decoder A/B (product vs variants) + the decoder's GPU tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
tag=${1:-r06dec}; shift
tools/gpu_dec_variants.sh "$@" > gpurun_out/${tag}_variants.log 2>&1; cat gpurun_out/${tag}_variants.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hints.py -x -q --timeout 300 > gpurun_out/${tag}_tests.log 2>&1; tail -5 gpurun_out/${tag}_tests.log
