"""Throw-away measurement: what do the entry-point table's rows cost the compressor?  (kernel ms with / without a table attached)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, lz4_amd
from bench import gen_data
nb, bs = 256, 4 << 20
ctx = lz4_amd.Context(0)
s = torch.cuda.current_stream().cuda_stream
for pct in (60, 90, 20):
    data = torch.from_numpy(gen_data(nb * bs, pct, 0)).cuda()
    hints = torch.zeros((nb, lz4_amd.hint_bytes(bs)), dtype=torch.uint8, device="cuda")
    _, cs, plan = lz4_amd.compress_blocks(ctx, data, bs)
    _, cs2, plan2 = lz4_amd.compress_blocks(ctx, data, bs, hints=hints)
    assert cs == cs2
    a = min(plan.launch_timed(s)[0][0] for _ in range(5)); b = min(plan2.launch_timed(s)[0][0] for _ in range(5))
    print("P%d: compress %.3f ms without a table, %.3f ms writing it (+%.1f %%)" % (pct, a, b, 100 * (b - a) / a))
