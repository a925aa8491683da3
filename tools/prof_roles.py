"""Developer aid: one wave's cycles per tile by phase in the paired compress kernel (library built with -DLZ4AMD_PROF_ROLES=<wave> [-DLZ4AMD_PROF_ROLES_SEL=0 measuring tiles | 1 writing tiles]). GPU only."""
import ctypes, os, sys, statistics
os.environ["LZ4AMD_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, lz4_amd
from bench import gen_data
nb, bs = 256, 4 << 20
pct = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ctx = lz4_amd.Context(0)
data = torch.from_numpy(gen_data(nb * bs, pct, 0)).cuda()
comp, csizes, plan = lz4_amd.compress_blocks(ctx, data, bs)
for _ in range(2):
    km, tot = plan.launch_timed(torch.cuda.current_stream().cuda_stream)
print("%s P%d: compress kernel ms %.3f" % (os.environ.get("LZ4AMD_LIB", "product"), pct, km[0]))
L = lz4_amd.lib()
w = (ctypes.c_ulonglong * (256 * 8))()
n = L.lz4amd_plan_profile(plan._h, w, len(w))
names = ["(settle,) probe + list", "wait for the partner's list, measure / select / records", "wait for the settle, write out strips", "wait until the tile may be inserted",
         "insert", "wait until the tile before may leave", "store the tile before", "wait at the barrier"]
tiles = bs // 8192
tot = 0
for k, name in enumerate(names):
    d = [w[i * 8 + k] for i in range(n // 8)]
    tot += statistics.median(d)
    print("  %-58s cycles per block: median %9d   per full tile of the block %6d" % (name, statistics.median(d), statistics.median(d) // tiles))
print("  sum per tile %d (a role's tiles are half of them)" % (tot // tiles))
