"""Throw-away measurement: compressed size of the emulated fast compressor against the real reference on the datagen grid.
usage: emu_ratio.py [emu .so]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_kernels_emulated import emu_compress
emu = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests/simt/libemu_kernels.so"))
ref = ctypes.CDLL(os.path.join(ROOT, "oracle/_ref/liblz4_ref.so"))
dg = ctypes.CDLL(os.path.join(ROOT, "tools/libdatagen.so"))
dg.lz4amd_datagen.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, ctypes.c_uint32]
def gen(n, pct, seed=0):
    buf = ctypes.create_string_buffer(max(n, 1)); dg.lz4amd_datagen(buf, n, pct / 100.0, 0.0, seed); return buf.raw[:n]
def refsize(d):
    cap = ref.LZ4_compressBound(len(d)); out = ctypes.create_string_buffer(cap)
    return ref.LZ4_compress_default(d, out, len(d), cap)
total = 4 << 20
for pct in (20, 50, 60, 90, 95):
    data = gen(total, pct, 0)
    row = []
    for bs in (65536, 262144, 4 << 20):
        blocks = [data[i:i + bs] for i in range(0, total, bs)]
        ours = sum(r for r, _ in emu_compress(emu, blocks))
        theirs = sum(refsize(b) for b in blocks)
        row.append("%7d: %+6.2f%%" % (bs, 100.0 * (ours - theirs) / theirs))
    print("P%-3d" % pct, "  ".join(row), flush=True)
