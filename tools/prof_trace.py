"""Developer aid: per-region time stamps of workgroup 0's copy waves (variant built with -DLZ4AMD_DEC_TRACE). GPU only.
usage: LZ4AMD_LIB=variants/liblz4_amd_<trace>.so prof_trace.py [n_blocks] [block_bytes] [P] [hc_level]"""
import ctypes, os, sys, statistics, collections
os.environ["LZ4AMD_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, lz4_amd
from bench import gen_data
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4 << 20
pct = int(sys.argv[3]) if len(sys.argv) > 3 else 60
hc = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx = lz4_amd.Context(0)
data = torch.from_numpy(gen_data(nb * bs, pct, 0)).cuda()
hints = torch.zeros((nb, lz4_amd.hint_bytes(bs)), dtype=torch.uint8, device="cuda")
comp, csizes, _ = lz4_amd.compress_blocks(ctx, data, bs, hc_level=(hc or None), hints=hints)
out, res, plan = lz4_amd.decompress_blocks(ctx, comp, csizes, bs, nb * bs, hints=hints)
assert torch.equal(out, data)
L = lz4_amd.lib()
nw = 256 * 8 + (4 << 20) // 8
w = (ctypes.c_ulonglong * nw)()
n = L.lz4amd_plan_profile(plan._h, w, nw)
grid = min(nb, ctx.cus)
base = grid * 8
cnt = min(w[base], (4 << 20) // 80 - 2)
recs = {}
for i in range(cnt):
    q = [w[base + 1 + 10 * i + j] for j in range(10)]
    R = q[0] & 0xFFFFFFFF
    recs[R] = dict(wave=(q[0] >> 32) & 255, np=(q[0] >> 40) & 255, sleeps=(q[0] >> 48) & 0xFFFF, t=q[1:9])
print("regions", len(recs))
t0 = min(r["t"][0] for r in recs.values())
rel = lambda x: x - t0 if x else 0
print("region wave np sleeps | start fp_done first_wake first_land last_land full done | fp land_wait fin")
keys = sorted(recs)
for R in keys[300:340]:
    r = recs[R]; t = r["t"]
    print("%5d w%-2d %3d %3d | %8d %8d %8d %8d %8d %8d %8d | %6d %6d %6d" % (R, r["wave"], r["np"], r["sleeps"], rel(t[0]), rel(t[1]), rel(t[2]), rel(t[3]), rel(t[4]), rel(t[5]), rel(t[6]),
          t[1] - t[0], (t[4] or t[1]) - t[1], t[6] - (t[4] or t[1])))
pend = [r for r in recs.values() if r["np"]]
print("regions with pending pieces: %d of %d" % (len(pend), len(recs)))
med = statistics.median
print("median first pass %d ; pending regions: fp_done -> first landing %d, -> last landing %d ; last landing -> done %d ; regions without: fp_done -> done %d" % (
    med([r["t"][1] - r["t"][0] for r in recs.values()]), med([r["t"][3] - r["t"][1] for r in pend if r["t"][3]]), med([r["t"][4] - r["t"][1] for r in pend if r["t"][4]]),
    med([r["t"][6] - r["t"][4] for r in pend if r["t"][4]]), med([r["t"][6] - r["t"][1] for r in recs.values() if not r["np"]])))
# what made a region's first landing possible?  the latest landing / first-pass end / completion of a region below it, before
trig = collections.Counter(); lagv = collections.defaultdict(list)
for R in keys:
    r = recs[R]
    if not r["t"][3]: continue
    T = r["t"][3]; best = None
    for d in range(1, 13):
        q = recs.get(R - d)
        if not q: continue
        for name, idx in (("fp", 1), ("land", 4), ("done", 6)):
            x = q["t"][idx]
            if x and x < T and (best is None or x > best[0]): best = (x, name, d)
    if best: trig[(best[1], best[2])] += 1; lagv[(best[1], best[2])].append(T - best[0])
print("latest event of a lower region before my first landing -> count, median lag")
for k, c in trig.most_common(12): print(k, c, med(lagv[k]))
dn = sorted(r["t"][6] for r in recs.values())
print("completions span %d cycles, mean gap %.0f" % (dn[-1] - dn[0], (dn[-1] - dn[0]) / max(1, len(dn) - 1)))
