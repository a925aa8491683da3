"""Throw-away check: compress + decompress n blocks of 4 MiB on the GPU, printing as it goes (python -u)."""
import os, sys, faulthandler
faulthandler.dump_traceback_later(12, exit=False)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, lz4_amd
from bench import gen_data
nb, bs = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 4 << 20
print("lib", os.environ.get("LZ4AMD_LIB", "product"), flush=True)
ctx = lz4_amd.Context(0)
data = torch.from_numpy(gen_data(nb * bs, 60, 0)).cuda()
print("data ready", flush=True)
comp, csizes, plan = lz4_amd.compress_blocks(ctx, data, bs)
print("compressed", sum(csizes), flush=True)
out, res, _ = lz4_amd.decompress_blocks(ctx, comp, csizes, bs, nb * bs)
print("round trip", bool(torch.equal(out, data)), flush=True)
