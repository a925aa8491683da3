"""throw-away: small-block compress rate and size against the reference (oracle/_ref) by compressibility"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, lz4_amd
from bench import gen_data
ref = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle", "_ref", "liblz4_ref.so"))
ctx = lz4_amd.Context(0)
total = 64 << 20
for pct in (20, 60, 90):
    host = gen_data(total, pct, 7)
    data = torch.from_numpy(host).cuda()
    hb = host.tobytes()
    for bs in (4096, 16384, 32768, 65536):
        comp, csizes, plan = lz4_amd.compress_blocks(ctx, data, bs)
        km = min(plan.launch_timed(torch.cuda.current_stream().cuda_stream)[0][0] for _ in range(3))
        nref = min(total // bs, 256); rs = 0
        dst = ctypes.create_string_buffer(bs + bs // 255 + 16)
        for i in range(nref):
            rs += ref.LZ4_compress_default(hb[i * bs:(i + 1) * bs], dst, bs, len(dst))
        print("P%d %6d-byte blocks: %.2f ms per GiB  ours/ref size %.4f" % (pct, bs, km * (1 << 30) / total, sum(csizes[:nref]) / rs))
