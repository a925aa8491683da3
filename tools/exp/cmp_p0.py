"""Developer aid: compress kernel time on incompressible data: datagen -P0 (skewed literal bytes) and uniform random bytes. GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, lz4_amd, numpy as np
from bench import gen_data
nb, bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 4 << 20
ctx = lz4_amd.Context(0)
s = torch.cuda.current_stream().cuda_stream
for name, host in (("datagen -P0", gen_data(nb * bs, 0, 1)), ("uniform random", np.random.default_rng(1).integers(0, 256, nb * bs, dtype=np.uint8)), ("datagen -P90", gen_data(nb * bs, 90, 1)), ("datagen -P20", gen_data(nb * bs, 20, 1)),
                   ("datagen -P10", gen_data(nb * bs, 10, 1)), ("datagen -P60", gen_data(nb * bs, 60, 1))):
    data = torch.from_numpy(host).cuda()
    comp, cs, plan = lz4_amd.compress_blocks(ctx, data, bs)
    best = min(plan.launch_timed(s)[0][0] for _ in range(5))
    print("%-16s kernel ms %.3f for %d MiB  -> %.1f GB/s per 256 CUs equivalent %.1f ; ratio %.4f" % (name, best, nb * bs >> 20, nb * bs / best / 1e6, nb * bs / best / 1e6 * 256 / min(nb, 256), nb * bs / sum(cs)))
