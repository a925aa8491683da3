"""Developer aid: phase breakdown of the decompress kernel (LZ4AMD_PROF stamps). GPU only."""
import ctypes, os, sys
os.environ["LZ4AMD_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, lz4_amd
from bench import gen_data
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4 << 20
ctx = lz4_amd.Context(0)
data = torch.from_numpy(gen_data(nb * bs, 60, 0)).cuda()
comp, csizes, _ = lz4_amd.compress_blocks(ctx, data, bs)
out, res, plan = lz4_amd.decompress_blocks(ctx, comp, csizes, bs, nb * bs)
for _ in range(3):
    km, tot = plan.launch_timed(torch.cuda.current_stream().cuda_stream)
print("decompress kernel ms", km[0], "GB/s out", nb * bs / km[0] / 1e6)
L = lz4_amd.lib()
w = (ctypes.c_ulonglong * (256 * 8))()
n = L.lz4amd_plan_profile(plan._h, w, len(w))
import statistics
nw = n // 8
pre = [w[i * 8 + 1] - w[i * 8 + 0] for i in range(nw)]
emit = [w[i * 8 + 2] for i in range(nw)]
copy = [w[i * 8 + 3] for i in range(nw)]
tot = [w[i * 8 + 4] - w[i * 8 + 0] for i in range(nw)]
for name, d in (("preparse", pre), ("stream.emit+load+index", emit), ("stream.copy", copy), ("total", tot)):
    print(name, "cycles median", statistics.median(d), "max", max(d))
print("preparse: count+scan", statistics.median([w[i*8+5] for i in range(nw)]), "walk", statistics.median([w[i*8+6] for i in range(nw)]), "fix", statistics.median([w[i*8+7] & ((1<<48)-1) for i in range(nw)]), "fix iterations", statistics.median([w[i*8+7] >> 48 for i in range(nw)]))
assert torch.equal(out, data)
