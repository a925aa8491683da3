"""Randomized round-trip stress on the GPU (tests/test_gpu_parity.py::test_randomized_round_trip_stress runs it for 25 s).
usage: stress_gpu.py [seconds] [seed]"""
import ctypes, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, lz4_amd
from bench import gen_data
import numpy as np

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = lz4_amd.Context(0)
ref = None
p = os.path.join(ROOT, "oracle", "_ref", "liblz4_ref.so")
if os.path.exists(p):
    ref = ctypes.CDLL(p)
pool = {pct: gen_data(24 << 20, pct, 7) for pct in (0, 20, 50, 60, 90)}
pool[100] = np.zeros(24 << 20, dtype=np.uint8)
pool[101] = np.frombuffer(bytes(range(256)) * (24 << 12), dtype=np.uint8)         # period 256
t_end = time.time() + budget
rounds = blocks = bytes_ = tabled = 0
while time.time() < t_end:
    nb = rng.choice((1, 2, 7, 64, 300))
    sizes = [rng.choice((1, 5, 13, 64, 100, 4095, 65536, 65537, 262144, 1 << 20, 4 << 20)) if rng.random() < 0.5 else rng.randint(1, 300000) for _ in range(nb)]
    if sum(sizes) > 600 << 20:
        continue
    chunks = []
    for s in sizes:
        src = pool[rng.choice(list(pool))]
        o = rng.randrange(0, len(src) - s)
        chunks.append(src[o:o + s])
    host = np.concatenate(chunks)
    data = torch.from_numpy(host).cuda()
    offs = np.concatenate(([0], np.cumsum(sizes)))[:-1]
    hc = rng.choice((None, None, 3, 9))
    bound = [lz4_amd.compress_bound(s) for s in sizes]
    coff = np.concatenate(([0], np.cumsum(bound)))[:-1]
    comp = torch.empty(int(sum(bound)) + 64, dtype=torch.uint8, device="cuda")
    tab = lz4_amd.BlockTable([data.data_ptr() + int(o) for o in offs], sizes, [comp.data_ptr() + int(o) for o in coff], bound)
    plan = lz4_amd.Plan(ctx, lz4_amd.OP_COMPRESS if hc is None else lz4_amd.OP_COMPRESS_HC, tab, level=hc or 0)
    hints = None
    if hc is None and rng.random() < 0.5:                           # entry-point tables (include/lz4amd.h): written here, used by the decoder below
        hstride = lz4_amd.hint_bytes(max(sizes))
        hints = torch.zeros((nb, hstride), dtype=torch.uint8, device="cuda")
        plan.attach_hints(hints.data_ptr(), hstride)
        if rng.random() < 0.3:
            plan.set_acceleration(2)
    st = torch.cuda.current_stream().cuda_stream
    plan.launch(st); cs = plan.results(st); plan.close()
    assert all(c > 0 for c in cs), ("compress failed", sizes, cs)
    out = torch.full((len(host) + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    dt = lz4_amd.BlockTable([comp.data_ptr() + int(o) for o in coff], cs, [out.data_ptr() + int(o) for o in offs], sizes)
    dp = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, dt)
    if hints is not None:
        dp.attach_hints(hints.data_ptr(), hints.stride(0))
    dp.launch(st); res = dp.results(st)
    if hints is not None:
        used, rejected = dp.hint_stats()
        # (a block whose sequences are too dense for the table's room - fewer than ~8 source bytes each - has no table: its header stays 0)
        none = int((hints[:, :4].cpu().numpy().view(np.uint32)[:, 0] == 0).sum()) if rejected == 0 and used != sum(1 for s in sizes if s > 0) else 0
        assert rejected == 0 and used + none == sum(1 for s in sizes if s > 0), ("tables", used, rejected, none, nb)
        assert none * 20 <= nb + 19, ("many blocks without a table", none, nb)
        tabled += used
    dp.close()
    assert res == sizes, ("decode sizes", [(i, r, s) for i, (r, s) in enumerate(zip(res, sizes)) if r != s][:5])
    assert torch.equal(out[:len(host)], data), "decode mismatch"
    assert bool((out[len(host):] == 0xEE).all())
    if ref is not None and rng.random() < 0.3:                      # a few blocks through the real reference decoder
        ch = comp.cpu().numpy()
        for i in rng.sample(range(nb), min(nb, 4)):
            dst = ctypes.create_string_buffer(sizes[i])
            r = ref.LZ4_decompress_safe(ch[int(coff[i]):int(coff[i]) + cs[i]].tobytes(), dst, cs[i], sizes[i])
            assert r == sizes[i] and dst.raw == chunks[i].tobytes(), ("reference decode", i)
    rounds += 1; blocks += nb; bytes_ += len(host)
print("stress ok: %d rounds, %d blocks (%d decoded from entry-point tables), %.1f GiB" % (rounds, blocks, tabled, bytes_ / 2**30))
