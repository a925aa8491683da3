#!/bin/bash
# one GPU visit: the table-making tests, then a short bench line
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_hints.py -x -q -m gpu > gpurun_out/mk_tests.log 2>&1; tail -5 gpurun_out/mk_tests.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/mk_bench.json 2> gpurun_out/mk_bench.err; tail -3 gpurun_out/mk_bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/mk_bench.json").read().strip().splitlines()[-1])
print(j["value"], j.get("decode_of_foreign_blocks"))
PY
