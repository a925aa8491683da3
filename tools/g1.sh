cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
( timeout 120 python tools/prof_dec.py 256 4194304 60 ) > gpurun_out/g1_v2.log 2>&1; tail -8 gpurun_out/g1_v2.log
( LZ4AMD_DEC=v1 timeout 120 python tools/prof_dec.py 256 4194304 60 ) > gpurun_out/g1_v1.log 2>&1; tail -2 gpurun_out/g1_v1.log
( timeout 120 python tools/prof_dec.py 2048 4194304 60 ) > gpurun_out/g1_v2_2048.log 2>&1; tail -6 gpurun_out/g1_v2_2048.log
( timeout 120 python tools/prof_dec.py 4096 262144 60 ) > gpurun_out/g1_v2_256k.log 2>&1; tail -6 gpurun_out/g1_v2_256k.log
( timeout 120 python tools/prof_dec.py 16384 65536 60 ) > gpurun_out/g1_v2_64k.log 2>&1; tail -6 gpurun_out/g1_v2_64k.log
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/g1_tests.log 2>&1; tail -5 gpurun_out/g1_tests.log
