#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 300 python bench.py --steps 10 --warmup 2 --no-hc --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step','compress_GBps','decompress_GBps')}); print([(k['kernel'],k['avg_ms']) for k in d['kernels']])"
timeout 300 python bench.py --steps 10 --warmup 2 --no-hc --no-extras --no-cpu-baseline --no-hints 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('no tables:', {k: d[k] for k in ('value','ms_per_step','compress_GBps','decompress_GBps')}); print([(k['kernel'],k['avg_ms']) for k in d['kernels']])"
