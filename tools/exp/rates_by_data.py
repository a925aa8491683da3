"""throw-away: compress and decompress kernel rates of 256 x 4 MiB blocks by compressibility (datagen -P), device resident"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, lz4_amd
from bench import gen_data
ctx = lz4_amd.Context(0)
nb, bs = 256, 4 << 20
s = torch.cuda.current_stream().cuda_stream
for pct in (0, 20, 40, 60, 80, 90, 95, 100):
    data = torch.from_numpy(gen_data(nb * bs, pct, 0)).cuda()
    comp, csizes, cplan = lz4_amd.compress_blocks(ctx, data, bs)
    out, res, dplan = lz4_amd.decompress_blocks(ctx, comp, csizes, bs, nb * bs)
    assert torch.equal(out, data)
    cms = min(cplan.launch_timed(s)[0][0] for _ in range(3)); dms = min(dplan.launch_timed(s)[0][0] for _ in range(3))
    U, C = nb * bs, sum(csizes)
    print("P%-3d ratio %6.3f  compress %6.2f ms (%5.0f GB/s)  decompress %6.2f ms (%5.0f GB/s; U+C %5.0f GB/s)" % (pct, U / C, cms, U / cms / 1e6, dms, U / dms / 1e6, (U + C) / dms / 1e6))
