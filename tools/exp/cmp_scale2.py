import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, lz4_amd, numpy as np
from bench import gen_data
bs = 4 << 20
ctx = lz4_amd.Context(0)
s = torch.cuda.current_stream().cuda_stream
for name, host in (("P0", gen_data(256 * bs, 0, 1)), ("random", np.random.default_rng(1).integers(0, 256, 256 * bs, dtype=np.uint8))):
    for a, b in ((0, 64), (64, 128), (128, 192), (192, 256), (0, 256), (0, 8), (64, 72)):
        data = torch.from_numpy(host[a * bs:b * bs]).cuda()
        comp, cs, plan = lz4_amd.compress_blocks(ctx, data, bs)
        best = min(plan.launch_timed(s)[0][0] for _ in range(4))
        print("%-6s blocks %3d..%3d: %.3f ms" % (name, a, b, best), flush=True)
