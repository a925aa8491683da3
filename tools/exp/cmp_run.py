"""Throw-away: run the compress kernel a few times (for rocprofv3 --pmc runs over variant builds).  usage: cmp_run.py [P]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, lz4_amd
from bench import gen_data
pct = int(sys.argv[1]) if len(sys.argv) > 1 else 60
nb, bs = 256, 4 << 20
ctx = lz4_amd.Context(0)
data = torch.from_numpy(gen_data(nb * bs, pct, 0)).cuda()
comp, csizes, plan = lz4_amd.compress_blocks(ctx, data, bs)
s = torch.cuda.current_stream().cuda_stream
ms = min(plan.launch_timed(s)[0][0] for _ in range(4))
print("P%d %s: %.3f ms" % (pct, os.environ.get("LZ4AMD_LIB", "product"), ms))
