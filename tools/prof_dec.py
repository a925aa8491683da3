"""Developer aid: role breakdown of the streaming decompress kernel (LZ4AMD_PROF stamps). GPU only.
usage: prof_dec.py [n_blocks] [block_bytes] [P] [hc_level]      (NOHINTS=1: decode without the compressor's entry-point tables)"""
import ctypes, os, sys, statistics
if not os.environ.get("NOPROF"): os.environ["LZ4AMD_PROF"] = "1"        # NOPROF=1: kernel time only, without the stamps
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
use_hints = not os.environ.get("NOHINTS")
hints = torch.zeros((nb, lz4_amd.hint_bytes(bs)), dtype=torch.uint8, device="cuda") if use_hints else None
comp, csizes, _ = lz4_amd.compress_blocks(ctx, data, bs, hc_level=(hc or None), hints=hints)
out, res, plan = lz4_amd.decompress_blocks(ctx, comp, csizes, bs, nb * bs, hints=hints)
best = 1e9
for _ in range(5):
    km, tot = plan.launch_timed(torch.cuda.current_stream().cuda_stream)
    best = min(best, km[0])
U, C = nb * bs, sum(csizes)
print("decoder %s: %d x %d B P%d%s  kernel ms %.3f  GB/s out %.1f  (U+C)/t %.1f GB/s = %.3f of 8 TB/s" % (
    "v5 " + ("with tables %s" % (plan.hint_stats(),) if use_hints else "no tables"), nb, bs, pct, " hc%d" % hc if hc else "", best, U / best / 1e6, (U + C) / best / 1e6, (U + C) / best / 1e6 / 8000))
assert os.environ.get("NOCHECK") or torch.equal(out, data), "decode mismatch"
if not os.environ.get("NOPROF"):
    L = lz4_amd.lib()
    w = (ctypes.c_ulonglong * (256 * 8))()
    n = L.lz4amd_plan_profile(plan._h, w, len(w))
    nw = n // 8
    med = lambda f: statistics.median([f(i) for i in range(nw)])
    print("cycles per workgroup pass (median over %d workgroups; last block each):" % nw)
    print("  block total        %10d" % med(lambda i: w[i * 8]))
    print("  pre-parse %d  (P1 %d  P2 %d  P3 %d  P4 %d  list %d  P5 %d)" % (med(lambda i: w[i * 8 + 1]), med(lambda i: w[i * 8 + 2] & 0xFFFFFFFF), med(lambda i: w[i * 8 + 2] >> 32),
          med(lambda i: w[i * 8 + 3] & 0xFFFFFFFF), med(lambda i: w[i * 8 + 3] >> 32), med(lambda i: w[i * 8 + 4] & 0xFFFFFFFF), med(lambda i: w[i * 8 + 4] >> 32)))
    if use_hints:
        print("  parser: batches %d  lanes per batch %.1f  steps %d (careful %d)  cycles waiting %d  walking %d" % (
            med(lambda i: w[i * 8 + 2] & 0xFFFFFFFF), med(lambda i: (w[i * 8 + 2] >> 32) / max(1, w[i * 8 + 2] & 0xFFFFFFFF)), med(lambda i: w[i * 8 + 3] & 0xFFFFFFFF),
            med(lambda i: w[i * 8 + 3] >> 32), med(lambda i: (w[i * 8 + 4] & 0xFFFFFFFF) << 4), med(lambda i: (w[i * 8 + 4] >> 32) << 4)))
    print("  copy wave 0: regions %d, with retry %d, retry iterations %d" % (med(lambda i: w[i * 8 + 5] >> 48), med(lambda i: (w[i * 8 + 5] >> 32) & 0xFFFF), med(lambda i: w[i * 8 + 5] & 0xFFFFFFFF)))
    print("  copy wave 0: wait records %d  wait lead %d  work %d  retry (sources in flight) %d" % (
        med(lambda i: w[i * 8 + 6] & 0xFFFFFFFF), med(lambda i: w[i * 8 + 6] >> 32), med(lambda i: w[i * 8 + 7] & 0xFFFFFFFF), med(lambda i: w[i * 8 + 7] >> 32)))
