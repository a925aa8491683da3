#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
tag=${1:-r4i}
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/${tag}_gputests.log 2>&1; tail -5 gpurun_out/${tag}_gputests.log
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench.json'))
print({k: d[k] for k in ('value','ms_per_step','compress_GBps','decompress_GBps','ratio','ratio_vs_reference')}); print(d['roofline']['frac'], d['roofline_decompress']['frac'], d['entry_point_tables']['blocks_decoded_from_their_table']); print(d['decode_of_foreign_blocks']); print(d['by_compressibility']); print(d['frame']); print(d.get('hc',{}).get('compress_GBps'), d.get('hc',{}).get('optimal_parse_level12'))" ; tail -3 gpurun_out/${tag}_bench.err
