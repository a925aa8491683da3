# developer aid: the linked-blocks matrix of DESIGN.md 3.3 (device resident, side by side vs the serial chain)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python tools/exp/linked_speed.py 1024 4096 60 2>&1 | grep -E "^serial|^side"
timeout 300 python tools/exp/linked_speed.py 1024 1024 60 2>&1 | grep -E "^serial|^side"
timeout 300 python tools/exp/linked_speed.py 256 256 60 2>&1 | grep -E "^serial|^side"
timeout 300 python tools/exp/linked_speed.py 256 64 60 2>&1 | grep -E "^serial|^side"
timeout 300 python tools/exp/linked_speed.py 1024 64 60 2>&1 | grep -E "^side"
for g in 4 8 32; do echo "group $g"; LZ4AMD_CHAIN_GROUP=$g timeout 300 python tools/exp/linked_speed.py 256 64 60 2>&1 | grep -E "^side"; done
timeout 300 python tools/exp/linked_speed.py 256 4096 20 2>&1 | grep -E "^serial|^side"
timeout 300 python tools/exp/linked_speed.py 256 4096 99 2>&1 | grep -E "^serial|^side"
timeout 300 python tools/exp/linked_speed.py 256 64 99 2>&1 | grep -E "^serial|^side"
