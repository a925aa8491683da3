for v in product rc rcns; do
  [ "$v" = product ] && unset LZ4AMD_LIB || export LZ4AMD_LIB=variants/liblz4_amd_$v.so
  echo "== $v serial"; LZ4AMD_CHAIN_SERIAL=1 timeout 100 python -m pytest tests/test_gpu_linked.py -x -q 2>&1 | grep -E "passed|failed|fault" | tail -2
  echo "== $v sbs"; timeout 200 python -m pytest tests/test_gpu_linked.py -x -q 2>&1 | grep -E "passed|failed|fault" | tail -2
  timeout 200 python tools/exp/linked_speed.py 1024 4096 60 2>&1 | grep -E "side|units|fault" | tail -2
done
unset LZ4AMD_LIB
timeout 200 python tools/exp/linked_speed.py 256 64 60 2>&1 | tail -2; timeout 200 python tools/exp/linked_speed.py 256 4096 99 2>&1 | tail -2
