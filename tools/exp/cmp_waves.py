"""throw-away: match + emit cycles of each of the 16 waves of the compress kernel (variant built with -DLZ4AMD_PROF_WAVES)"""
import ctypes, os, sys, statistics
os.environ["LZ4AMD_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, lz4_amd
from bench import gen_data
nb, bs = 256, 4 << 20
ctx = lz4_amd.Context(0)
data = torch.from_numpy(gen_data(nb * bs, int(sys.argv[1]) if len(sys.argv) > 1 else 60, 0)).cuda()
comp, csizes, plan = lz4_amd.compress_blocks(ctx, data, bs)
for _ in range(3): km, tot = plan.launch_timed(torch.cuda.current_stream().cuda_stream)
L = lz4_amd.lib(); w = (ctypes.c_ulonglong * (256 * 8))(); n = L.lz4amd_plan_profile(plan._h, w, len(w))
per = [[] for _ in range(16)]
for g in range(n // 8):
    for i in range(8):
        v = w[g * 8 + i]; per[2 * i].append((v & 0xFFFFFFFF) << 4); per[2 * i + 1].append((v >> 32) << 4)
print("kernel ms", km[0])
print("match + emit cycles per wave (median over workgroups):", [int(statistics.median(p)) // 1000 for p in per], "K")
