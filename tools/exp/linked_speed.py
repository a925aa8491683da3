"""Device-resident decode of linked blocks: side by side against the serial chain (LZ4AMD_CHAIN_SERIAL=1).  usage: linked_speed.py [MiB] [block KiB] [P]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench, lz4_amd
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bk = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
pct = int(sys.argv[3]) if len(sys.argv) > 3 else 60
bs, nb = bk << 10, (mib << 20) // (bk << 10)
ctx = lz4_amd.Context(0)
src = torch.from_numpy(bench.gen_data(mib << 20, pct, 0)).cuda()
bound = lz4_amd.compress_bound(bs)
dst = torch.zeros(nb * bound + 64, dtype=torch.uint8, device="cuda")
table = lz4_amd.BlockTable([src.data_ptr() + i * bs for i in range(nb)], [bs] * nb, [dst.data_ptr() + i * bound for i in range(nb)], [bound] * nb)
plan = lz4_amd.Plan.compress_with_history(ctx, table, [min(i * bs, 65536) for i in range(nb)])
s = torch.cuda.current_stream().cuda_stream
plan.launch(s); cs = plan.results(s); plan.close()
out = torch.zeros(nb * bs + 64, dtype=torch.uint8, device="cuda")
for mode in ("serial", "side by side"):
    os.environ.pop("LZ4AMD_CHAIN_SERIAL", None)
    if mode == "serial": os.environ["LZ4AMD_CHAIN_SERIAL"] = "1"
    t0 = time.time()
    dplan = lz4_amd.Plan.chained(ctx, [dst.data_ptr() + i * bound for i in range(nb)], cs, out.data_ptr(), [bs] * nb)
    tc = time.time() - t0
    out.zero_()
    best = 1e9
    for _ in range(3):
        best = min(best, dplan.launch_timed(s)[1])
    res = dplan.results(s)
    ok = res == [bs] * nb and torch.equal(out[:nb * bs], src[:nb * bs])
    print("%-13s %d x %d KiB P%d: %.3f ms  %.1f GB/s out  plan %.1f ms  ratio %.3f  exact %s" % (mode, nb, bk, pct, best, nb * bs / best / 1e6, tc * 1e3, nb * bs / sum(cs), ok))
    try: print("   ", dplan.chain_stats())
    except Exception as e: print("   ", e)
    dplan.close()
