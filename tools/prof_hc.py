"""Developer aid: timing and phase breakdown of the HC kernel (LZ4AMD_PROF cycle counts). GPU only."""
import ctypes, os, sys, statistics
os.environ["LZ4AMD_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, lz4_amd
from bench import gen_data
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256 << 10
pct = int(sys.argv[3]) if len(sys.argv) > 3 else 60
levels = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [9]      # several levels: "9,2,12"
ctx = lz4_amd.Context(0)
data = torch.from_numpy(gen_data(nb * bs, pct, 0)).cuda()
s = torch.cuda.current_stream().cuda_stream
L = lz4_amd.lib()
for level in levels:
  comp, csizes, plan = lz4_amd.compress_blocks(ctx, data, bs, hc_level=level)
  runs = 2
  for _ in range(runs):
    km, tot = plan.launch_timed(s)
  print("HC level %d, %d x %d B P%d: kernel ms %.2f  GB/s in %.2f  ratio %.3f" % (level, nb, bs, pct, km[0], nb * bs / km[0] / 1e6, nb * bs / sum(csizes)))
  w = (ctypes.c_ulonglong * (256 * 8))()
  n = L.lz4amd_plan_profile(plan._h, w, len(w))
  names = ["chain build", "search band 0", "search band 1", "optimal parse: forward pass of wave 0 (levels 10-12)", "parse", "offsets + emit"]
  blocks_per_wg = nb / (n // 8) * (runs + 1)
  for k, name in enumerate(names):
    d = [w[i * 8 + k] for i in range(n // 8)]
    print("%-16s cycles per block: median %.0f  max %.0f" % (name, statistics.median(d) / blocks_per_wg, max(d) / blocks_per_wg))

  if "hcprof" in os.environ.get("LZ4AMD_LIB", ""):     # tools/build_variant.sh hcprof -DLZ4AMD_PROF_HC
    med = lambda f: statistics.median([f(i) for i in range(n // 8)]) / blocks_per_wg
    trips = med(lambda i: w[i * 8 + 3] & 0xFFFFFFFF); lanes = med(lambda i: w[i * 8 + 3] >> 32)
    hits = med(lambda i: w[i * 8 + 6] & 0xFFFFFFFF); hlanes = med(lambda i: w[i * 8 + 6] >> 32)
    loop = med(lambda i: (w[i * 8 + 7] & 0xFFFFFFFF) << 4); wait = med(lambda i: (w[i * 8 + 7] >> 32) << 4)
    print("band 0, wave 0, per block: loop trips %.0f, active lanes per trip %.1f; measure trips %.0f, lanes per measure trip %.1f; cycles in the loop %.0f, at the tile's barrier %.0f" % (
        trips, lanes / max(trips, 1), hits, hlanes / max(hits, 1), loop, wait))
