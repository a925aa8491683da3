"""Developer aid: random chains of linked blocks through lz4amd_plan_create_decompress_chained under every scheme (the serial chain, side by side with twins,
with the second launch from tables, with neither), against the source and against each other.  usage: stress_linked.py [seconds] [seed]     GPU only."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, lz4_amd
from bench import gen_data

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
ctx = lz4_amd.Context(0)
S = torch.cuda.current_stream().cuda_stream


def make_data(n):
    out = bytearray()
    while len(out) < n:
        k = rng.choice(("gen", "gen", "gen", "noise", "run", "period", "back"))
        m = rng.choice((100, 3000, 70000, 300000, 1500000))
        if k == "gen": out += gen_data(m, rng.choice((20, 50, 60, 90, 99)), rng.randrange(1000)).tobytes()
        elif k == "noise": out += rng.randbytes(min(m, 200000))
        elif k == "run": out += bytes([rng.randrange(256)]) * m
        elif k == "period": out += (rng.randbytes(rng.choice((2, 7, 255, 4000))) * (m // 2 + 1))[:m]
        elif len(out) > 70000:                                   # a copy from up to 64 KB back (far offsets at any position of a unit)
            d = rng.choice((65535, 65530, 65300, 40000, 1000)); L = rng.choice((40, 500, 5000))
            for i in range(L): out.append(out[len(out) - d])
    return bytes(out[:n])


def chain(data, sizes, cap):
    whole = torch.frombuffer(bytearray(data + b"\0" * 64), dtype=torch.uint8).cuda()
    bound = lz4_amd.compress_bound(cap)
    dst = torch.zeros(len(sizes) * bound + 64, dtype=torch.uint8, device="cuda")
    offs = [sum(sizes[:i]) for i in range(len(sizes))]
    table = lz4_amd.BlockTable([whole.data_ptr() + o for o in offs], sizes, [dst.data_ptr() + i * bound for i in range(len(sizes))], [bound] * len(sizes))
    plan = lz4_amd.Plan.compress_with_history(ctx, table, [min(o, 65536) for o in offs])
    plan.launch(S); cs = plan.results(S); plan.close()
    assert all(c > 0 for c in cs)
    return dst, bound, cs


def decode(dst, bound, cs, cap, total, env):
    out = torch.full((total + cap + 128,), 0xEE, dtype=torch.uint8, device="cuda")
    for k in ("LZ4AMD_CHAIN_SERIAL", "LZ4AMD_CHAIN_TWINS", "LZ4AMD_CHAIN_TABLES"): os.environ.pop(k, None)
    os.environ.update(env)
    try:
        plan = lz4_amd.Plan.chained(ctx, [dst.data_ptr() + i * bound for i in range(len(cs))], cs, out.data_ptr() + 16, [cap] * len(cs))
    finally:
        for k in env: os.environ.pop(k, None)
    for _ in range(2):
        plan.launch(S); res = plan.results(S)
    plan.close()
    return res, out


t0, rounds, blocks, nbytes = time.time(), 0, 0, 0
modes = [{"LZ4AMD_CHAIN_SERIAL": "1"}, {"LZ4AMD_CHAIN_TWINS": "1", "LZ4AMD_CHAIN_TABLES": "0"}, {"LZ4AMD_CHAIN_TWINS": "0", "LZ4AMD_CHAIN_TABLES": "1"},
         {"LZ4AMD_CHAIN_TWINS": "0", "LZ4AMD_CHAIN_TABLES": "0"}, {}]
while time.time() - t0 < secs:
    cap = rng.choice((4096, 65536, 65536, 262144, 1 << 20, 4 << 20))
    n = rng.randrange(2, 40 if cap >= (1 << 20) else 200)
    ragged = rng.random() < 0.4
    sizes = [rng.randrange(1, cap + 1) if ragged and rng.random() < 0.3 else cap for _ in range(n)]
    if sum(sizes) > (96 << 20): sizes = sizes[:max(2, (96 << 20) // cap)]
    data = make_data(sum(sizes))
    dst, bound, cs = chain(data, sizes, cap)
    bad = rng.random() < 0.15
    if bad:                                                      # a damaged block: the chain ends there, the blocks before it stand
        k = rng.randrange(len(cs)); cs = list(cs); cs[k] = max(1, cs[k] - rng.randrange(1, 9))
    ref = None
    for env in modes:
        res, out = decode(dst, bound, cs, cap, len(data), env)
        good = len(res) if all(r >= 0 for r in res) else [i for i, r in enumerate(res) if r < 0][0]
        got = out[16:16 + sum(sizes[:good])].cpu().numpy().tobytes()
        assert bytes(out[:16].cpu().numpy()) == b"\xEE" * 16, (seed, rounds, env)
        if ref is None:
            ref = (res[:good], good)
            if not bad: assert good == len(sizes) and res == sizes, (seed, rounds, env, res[:8])
        assert (res[:good], good) == ref and all(r < 0 for r in res[good:]), (seed, rounds, env, good, ref[1])
        assert got == data[:sum(sizes[:good])], (seed, rounds, env, cap, len(sizes))
    rounds += 1; blocks += len(sizes); nbytes += len(data)
print("linked stress ok: %d chains, %d blocks, %.1f MiB, every chain under %d schemes (seed %d)" % (rounds, blocks, nbytes / 2**20, len(modes), seed))
