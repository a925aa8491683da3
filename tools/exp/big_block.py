import sys
sys.path.insert(0, "/root/repo")
import torch, lz4_amd
from bench import gen_data
ctx = lz4_amd.Context(0)
for n, pct in ((512 << 20, 60), (300 * 1000 * 1000 + 7, 95), (64 << 20, 0)):
    data = torch.from_numpy(gen_data(n, pct, 5)).cuda()
    comp, cs, _ = lz4_amd.compress_blocks(ctx, data, n)
    out, res, _ = lz4_amd.decompress_blocks(ctx, comp, cs, n, n)
    print(n, pct, cs, res, bool(torch.equal(out, data)))
