"""Developer aid: decode of blocks the reference compressed (LZ4_compress_default on the host; 32 blocks, 8 times over), without tables - the launch rocprofv3 profiles for the foreign-input rows of profiles/. GPU only.  usage: prof_refdec.py [P]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, lz4_amd
from bench import gen_data, reference_blocks
nb, bs = 256, 4 << 20
pct = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ctx = lz4_amd.Context(0)
s = torch.cuda.current_stream().cuda_stream
host = gen_data(32 * bs, pct, 0)
rcomp, rsz = reference_blocks(host, bs, 32)
rdev = torch.from_numpy(rcomp).cuda()
out = torch.empty(nb * bs, dtype=torch.uint8, device="cuda")
rtab = lz4_amd.BlockTable([rdev.data_ptr() + (i % 32) * rcomp.shape[1] for i in range(nb)], [rsz[i % 32] for i in range(nb)], [out.data_ptr() + i * bs for i in range(nb)], [bs] * nb)
rplan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, rtab)
rplan.launch(s)
assert rplan.results(s) == [bs] * nb and torch.equal(out[:32 * bs], torch.from_numpy(host).cuda())
ms = min(rplan.launch_timed(s)[0][0] for _ in range(5))
print("reference-compressed P%d blocks, no tables: %.3f ms  %.1f GB/s" % (pct, ms, nb * bs / ms / 1e6))
