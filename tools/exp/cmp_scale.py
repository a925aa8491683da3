"""Developer aid: compress kernel time against the number of blocks (workgroups busy) and the compressibility. GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, lz4_amd, numpy as np
from bench import gen_data
bs = 4 << 20
ctx = lz4_amd.Context(0)
s = torch.cuda.current_stream().cuda_stream
for pct in (0, 2, 5, 10, 60):
    host = gen_data(256 * bs, pct, 1)
    for nb in (64, 128, 192, 256):
        data = torch.from_numpy(host[:nb * bs]).cuda()
        comp, cs, plan = lz4_amd.compress_blocks(ctx, data, bs)
        best = min(plan.launch_timed(s)[0][0] for _ in range(4))
        print("P%-2d %3d blocks: %.3f ms  ratio %.4f" % (pct, nb, best, nb * bs / sum(cs)), flush=True)
