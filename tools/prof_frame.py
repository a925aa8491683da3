"""Developer aid: the device-resident configs[2] object of bench.py on its own (compress with history, XXH32 of the blocks, the gather
into frame layout, the chained decode of the linked blocks) - the launches rocprofv3 profiles for the frame rows of profiles/. GPU only."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, lz4_amd
from bench import gen_data, bench_frame_device, stream_copy_gbps
nb, bs = 256, 4 << 20
ctx = lz4_amd.Context(0)
s = torch.cuda.current_stream().cuda_stream
data = torch.from_numpy(gen_data(nb * bs, 60, 0)).cuda()
out = torch.empty_like(data)
r = bench_frame_device(ctx, lz4_amd, torch, data, out, s, bs, stream_copy_gbps(ctx, lz4_amd, torch, 1 << 30, s))
print(json.dumps({k: v for k, v in r.items() if not isinstance(v, dict)}))
