"""Developer aid: decode reference-compressed 4 MiB blocks with and without lz4amd_plan_make_hints, print per-block results."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, lz4_amd
import test_gpu_hints as tg, test_hints_emulated as th
L = ctypes.CDLL(os.path.join(ROOT, "tools", "libdatagen.so"))
L.lz4amd_datagen.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, ctypes.c_uint32]
R = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "liblz4_ref.so"))
def gen(n, pct, seed):
    b = ctypes.create_string_buffer(n); L.lz4amd_datagen(b, n, pct / 100.0, 0.0, seed); return b.raw[:n]
ctx = lz4_amd.Context(0)
cases = []
for spec in ((4 << 20, 60, 7), (4 << 20, 20, 8), (4 << 20, 90, 9), (1 << 20, 60, 3), (300000, 90, 4)):
    d = gen(*spec); cap = len(d) + len(d) // 255 + 16; cb = ctypes.create_string_buffer(cap)
    n = R.LZ4_compress_default(d, cb, len(d), cap)
    cases.append((d, cb.raw[:n]))
blocks = [c for _, c in cases]; wants = [d for d, _ in cases]
for mode in ("plain", "make", "make-alone"):
    for sel in ([list(range(len(cases)))] if mode != "make-alone" else [[i] for i in range(len(cases))]):
        bl = [blocks[i] for i in sel]; wa = [wants[i] for i in sel]
        empty = [bytes(th.hint_bytes(len(d))) for d in wa]
        made = [] if mode != "plain" else None
        outs, used, rej = tg.gpu_decompress_tables(ctx, bl, [len(d) for d in wa], empty, make=made)
        print(mode, sel, [(r, o == d) for d, (r, o) in zip(wa, outs)], used, rej, made[1] if made else None, flush=True)
