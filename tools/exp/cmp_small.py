"""Throw-away measurement: compress kernel time on small blocks (16384 x 64 KiB, 4096 x 256 KiB of datagen -P60)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, lz4_amd
from bench import gen_data
ctx = lz4_amd.Context(0)
s = torch.cuda.current_stream().cuda_stream
data = torch.from_numpy(gen_data(1 << 30, 60, 0)).cuda()
for bs in (65536, 262144, 32768):
    comp, cs, plan = lz4_amd.compress_blocks(ctx, data, bs)
    ms = min(plan.launch_timed(s)[0][0] for _ in range(3))
    print("%s: %d x %d B: compress %.3f ms  %.1f GB/s  ratio %.4f" % (os.environ.get("LZ4AMD_LIB", "product"), data.numel() // bs, bs, ms, data.numel() / ms / 1e6, data.numel() / sum(cs)))
