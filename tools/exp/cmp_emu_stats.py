"""Developer tool: trip counts of the fast compressor's parse / emit loops on the CPU interpreter.
build: cd tests/simt && g++ -O2 -std=c++17 -shared -fPIC -DLZ4AMD_EMU_STATS -o /tmp/libemu_stats.so emu_kernels.cpp simt_emu.cpp -lpthread
usage: cmp_emu_stats.py /tmp/libemu_stats.so [pct] [MiB]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_kernels_emulated import emu_compress
emu = ctypes.CDLL(sys.argv[1])
pct = int(sys.argv[2]) if len(sys.argv) > 2 else 60
mib = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dg = ctypes.CDLL(os.path.join(ROOT, "tools/libdatagen.so"))
dg.lz4amd_datagen.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, ctypes.c_uint32]
n = mib << 20
buf = ctypes.create_string_buffer(n); dg.lz4amd_datagen(buf, n, pct / 100.0, 0.0, 0)
res = emu_compress(emu, [buf.raw[:n]])
st = list((ctypes.c_ulonglong * 16).in_dll(emu, "lz4amd_emu_stats"))
tiles = n / 8192
names = ["parse passes", "runs measured", "select repeats (long match)", "long-match extra trips", "records", "list_round extra trips", "probe_list calls",
         "emit strips", "emit record passes", "emit literal passes", "pair strips (both lists)", "  their runs", "  their passes (ceil runs/64)"]
print("P%d %d MiB: %d bytes; per 8 KB tile:" % (pct, mib, res[0][0]))
for k, nm in enumerate(names): print("  %-34s %8.2f" % (nm, st[k] / tiles))
