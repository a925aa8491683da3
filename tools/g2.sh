cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
( timeout 120 python tools/prof_dec.py 256 4194304 60 ) > gpurun_out/g2_v2.log 2>&1; tail -8 gpurun_out/g2_v2.log
