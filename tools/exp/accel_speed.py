"""Throw-away measurement: compress kernel time and ratio by acceleration (256 x 4 MiB datagen).  usage: accel_speed.py [P]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, lz4_amd
from bench import gen_data
pct = int(sys.argv[1]) if len(sys.argv) > 1 else 60
nb, bs = 256, 4 << 20
ctx = lz4_amd.Context(0)
data = torch.from_numpy(gen_data(nb * bs, pct, 0)).cuda()
stride = (lz4_amd.compress_bound(bs) + 255) & ~255
comp = torch.empty((nb, stride), dtype=torch.uint8, device="cuda")
tab = lz4_amd.BlockTable([data.data_ptr() + i * bs for i in range(nb)], [bs] * nb, [comp.data_ptr() + i * stride for i in range(nb)], [stride] * nb)
s = torch.cuda.current_stream().cuda_stream
for accel in (1, 2):
    plan = lz4_amd.Plan(ctx, lz4_amd.OP_COMPRESS, tab)
    plan.set_acceleration(accel)
    plan.launch(s); cs = plan.results(s)
    ms = min(plan.launch_timed(s)[0][0] for _ in range(5))
    print("P%d acceleration %d: %.3f ms per GiB, %.1f GB/s, ratio %.4f" % (pct, accel, ms, nb * bs / ms / 1e6, nb * bs / sum(cs)))
