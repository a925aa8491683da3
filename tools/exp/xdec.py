"""Developer aid: blocks compressed by one build, decoded (without tables) by another.  usage: xdec.py save|load <file>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, lz4_amd
from bench import gen_data
nb, bs = 256, 4 << 20
ctx = lz4_amd.Context(0)
s = torch.cuda.current_stream().cuda_stream
data = torch.from_numpy(gen_data(nb * bs, 60, 0)).cuda()
if sys.argv[1] == "save":
    comp, csizes, _ = lz4_amd.compress_blocks(ctx, data, bs)
    torch.save({"comp": comp.cpu(), "cs": csizes}, sys.argv[2])
else:
    d = torch.load(sys.argv[2]); comp = d["comp"].cuda(); csizes = d["cs"]
    out, res, plan = lz4_amd.decompress_blocks(ctx, comp, csizes, bs, nb * bs)
    assert torch.equal(out, data)
    print("decode (no tables) of %s by %s: %.3f ms" % (sys.argv[2], os.environ.get("LZ4AMD_LIB", "product"), min(plan.launch_timed(s)[0][0] for _ in range(5))))
