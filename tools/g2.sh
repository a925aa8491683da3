cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
( LZ4AMD_DEC=v2 timeout 120 python tools/prof_dec.py 256 4194304 60 ) > gpurun_out/g2_v2.log 2>&1; tail -6 gpurun_out/g2_v2.log
( LZ4AMD_DEC=v2 timeout 120 python tools/prof_dec.py 4096 262144 60 9 ) > gpurun_out/g2_v2hc.log 2>&1; tail -6 gpurun_out/g2_v2hc.log
