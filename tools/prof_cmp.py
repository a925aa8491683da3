"""Developer aid: phase breakdown of the compress kernel (LZ4AMD_PROF cycle counts). GPU only.
usage: prof_cmp.py [n_blocks] [P] [acceleration]   (LZ4AMD_LIB=variants/liblz4_amd_match.so, built with -DLZ4AMD_PROF_MATCH: wave 0's match phases)"""
import ctypes, os, sys, statistics
os.environ["LZ4AMD_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, lz4_amd
from bench import gen_data
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bs = int(os.environ.get("BS", 4 << 20))
pct = int(sys.argv[2]) if len(sys.argv) > 2 else 60
accel = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = lz4_amd.Context(0)
data = torch.from_numpy(gen_data(nb * bs, pct, 0)).cuda()
comp, csizes, plan = lz4_amd.compress_blocks(ctx, data, bs, acceleration=accel)
for _ in range(3):
    km, tot = plan.launch_timed(torch.cuda.current_stream().cuda_stream)
print("P%d acceleration %d: compress kernel ms %.3f  GB/s in %.1f  ratio %.4f" % (pct, accel, km[0], nb * bs / km[0] / 1e6, nb * bs / sum(csizes)))
L = lz4_amd.lib()
w = (ctypes.c_ulonglong * (256 * 8))()
n = L.lz4amd_plan_profile(plan._h, w, len(w))
if "match" in os.environ.get("LZ4AMD_LIB", ""):
    names = ["records", "(emit, wave 0 >> 4)", "probe + runs + list", "-", "-", "list fence", "measure", "select"]
else:
    names = ["loop top (prefetch issue, barrier)", "settle + match (wave 0)", "barrier wait after emit", "insert + flush", "emit (wave 0)", "match+emit: slowest wave", "match+emit: fastest wave", "match+emit: mean wave"]
tiles = max(1, bs // int(os.environ.get("TILE", 8192)))
for k, name in enumerate(names):
    d = [w[i * 8 + k] for i in range(n // 8)]
    print("  %-40s cycles per block: median %10d  max %10d   per tile %7d" % (name, statistics.median(d), max(d), statistics.median(d) // tiles))
