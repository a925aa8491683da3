"""Developer aid: phase breakdown of the decompress kernel (LZ4AMD_PROF stamps). GPU only."""
import ctypes, os, sys
os.environ["LZ4AMD_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, lz4_amd
from bench import gen_data
nb, bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 4 << 20
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
names = ["walk", "fix", "scan+emit", "copy"]
for k in range(4):
    d = [w[i * 8 + k + 1] - w[i * 8 + k] for i in range(n // 8)]
    print(names[k], "cycles median", statistics.median(d), "max", max(d))
print("nseq", w[5], "total", w[6], "csize", w[7])
assert torch.equal(out, data)
