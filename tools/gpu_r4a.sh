#!/bin/bash
# round 4, first visit: the entry-point tables on the real machine - tests, decoder shapes with / without tables, bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
tag=${1:-r4a}
timeout 300 python -m pytest tests/test_gpu_hints.py -m gpu -x -q --timeout 120 > gpurun_out/${tag}_hinttests.log 2>&1; tail -5 gpurun_out/${tag}_hinttests.log
for shape in "256 4194304 60" "256 4194304 20" "256 4194304 90" "4096 262144 60" "16384 65536 60"; do
  ( NOPROF=1 timeout 120 python tools/prof_dec.py $shape ) 2>&1 | grep -E "^decoder|Error|error" 
  ( NOPROF=1 NOHINTS=1 timeout 120 python tools/prof_dec.py $shape ) 2>&1 | grep -E "^decoder|Error|error"
done
( timeout 120 python tools/prof_dec.py 256 4194304 60 ) > gpurun_out/${tag}_profdec60.log 2>&1; tail -7 gpurun_out/${tag}_profdec60.log
timeout 400 python -m pytest tests -m gpu -x -q --timeout 120 > gpurun_out/${tag}_gputests.log 2>&1; tail -3 gpurun_out/${tag}_gputests.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-hc > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench.json'))
print({k: d[k] for k in ('value','ms_per_step','compress_GBps','decompress_GBps','ratio')}); print(d['roofline_decompress']['frac'], d['entry_point_tables']); print(d['decode_of_foreign_blocks'])" ; tail -3 gpurun_out/${tag}_bench.err
