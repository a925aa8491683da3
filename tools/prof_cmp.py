"""Developer aid: phase breakdown of the compress kernel (LZ4AMD_PROF cycle counts). GPU only."""
import ctypes, os, sys, statistics
os.environ["LZ4AMD_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, lz4_amd
from bench import gen_data
nb, bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 4 << 20
ctx = lz4_amd.Context(0)
data = torch.from_numpy(gen_data(nb * bs, 60, 0)).cuda()
comp, csizes, plan = lz4_amd.compress_blocks(ctx, data, bs)
for _ in range(3):
    km, tot = plan.launch_timed(torch.cuda.current_stream().cuda_stream)
print("compress kernel ms", km[0], "GB/s in", nb * bs / km[0] / 1e6, "ratio", nb * bs / sum(csizes))
L = lz4_amd.lib()
w = (ctypes.c_ulonglong * (256 * 8))()
n = L.lz4amd_plan_profile(plan._h, w, len(w))
names = ["wait ring/prefetch issue", "match (wave 0)", "match barrier wait", "offsets + insert", "emit (wave 0)", "match+emit: slowest wave (sum over tiles)", "match+emit: fastest wave", "match+emit: mean wave"]
for k, name in enumerate(names):
    d = [w[i * 8 + k] for i in range(n // 8)]
    print(name, "cycles median", statistics.median(d), "max", max(d))
