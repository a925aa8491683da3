"""Developer aid: where the decoder's first parser wave spends its time (variant build -DLZ4AMD_PROF_PARSER, tools/build_variant.sh pp).
usage: LZ4AMD_LIB=variants/liblz4_amd_pp.so python tools/prof_parser.py [n_blocks] [block_bytes] [P]"""
import ctypes, os, sys, statistics
os.environ["LZ4AMD_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, lz4_amd
from bench import gen_data
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4 << 20
pct = int(sys.argv[3]) if len(sys.argv) > 3 else 60
ctx = lz4_amd.Context(0)
data = torch.from_numpy(gen_data(nb * bs, pct, 0)).cuda()
hints = torch.zeros((nb, lz4_amd.hint_bytes(bs)), dtype=torch.uint8, device="cuda")
comp, csizes, _ = lz4_amd.compress_blocks(ctx, data, bs, hints=hints)
out, res, plan = lz4_amd.decompress_blocks(ctx, comp, csizes, bs, nb * bs, hints=hints)
best = min(plan.launch_timed(torch.cuda.current_stream().cuda_stream)[0][0] for _ in range(5))
assert torch.equal(out, data)
w = (ctypes.c_ulonglong * (256 * 8))()
n = lz4_amd.lib().lz4amd_plan_profile(plan._h, w, len(w)); nw = n // 8
med = lambda f: statistics.median([f(i) for i in range(nw)])
print("P%d %d x %d: kernel ms %.3f; block total %d cycles" % (pct, nb, bs, best, med(lambda i: w[i * 8])))
print("  parser wave 0: batches %d  lanes/batch %.1f  steps %d  | wait (all) %d  walk %d" % (med(lambda i: w[i*8+2] & 0xFFFFFFFF), med(lambda i: (w[i*8+2] >> 32) / max(1, w[i*8+2] & 0xFFFFFFFF)),
      med(lambda i: w[i*8+3] & 0xFFFFFFFF), med(lambda i: (w[i*8+4] & 0xFFFFFFFF) << 4), med(lambda i: (w[i*8+4] >> 32) << 4)))
print("  lock %d  selection (+ no-work sleeps: %d) %d  walk + wait for turn %d  publish %d" % (med(lambda i: (w[i*8+5] & 0xFFFFFFFF) << 4), med(lambda i: w[i*8+7]),
      med(lambda i: (w[i*8+5] >> 32) << 4), med(lambda i: (w[i*8+6] & 0xFFFFFFFF) << 4), med(lambda i: (w[i*8+6] >> 32) << 4)))
