"""throw-away: compress kernel rate by block size (device resident)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, lz4_amd
from bench import gen_data
ctx = lz4_amd.Context(0)
total = 1 << 30
data = torch.from_numpy(gen_data(total, 60, 0)).cuda()
for bs in (16384, 65536, 262144, 1 << 20, 4 << 20):
    comp, csizes, plan = lz4_amd.compress_blocks(ctx, data, bs)
    km = min(plan.launch_timed(torch.cuda.current_stream().cuda_stream)[0][0] for _ in range(3))
    print("compress %7d-byte blocks x %5d: %.3f ms per GiB, %.1f GB/s, ratio %.3f" % (bs, total // bs, km, total / km / 1e6, total / sum(csizes)))
