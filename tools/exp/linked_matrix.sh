timeout 600 python -m pytest tests/test_gpu_linked.py -x -q 2>&1 | tail -3
timeout 300 python tools/exp/linked_speed.py 1024 4096 60 2>&1 | tail -1
timeout 300 python tools/exp/linked_speed.py 256 64 60 2>&1 | tail -1
LZ4AMD_CHAIN_GROUP=4 timeout 300 python tools/exp/linked_speed.py 256 64 60 2>&1 | tail -1
LZ4AMD_CHAIN_GROUP=1 timeout 300 python tools/exp/linked_speed.py 256 64 60 2>&1 | tail -1
timeout 300 python tools/exp/linked_speed.py 256 64 99 2>&1 | tail -2
timeout 300 python tools/exp/linked_speed.py 256 4096 99 2>&1 | tail -2
timeout 300 python tools/exp/linked_speed.py 256 4096 20 2>&1 | tail -2
