"""Developer tool: trip counts of the HC search on the CPU interpreter (build with -DLZ4AMD_EMU_STATS, see the bottom).
usage: hc_emu_stats.py <emu .so built with -DLZ4AMD_EMU_STATS> [pct] [level] [nblocks]
Counters per block: band 0 {walks, kept without a walk, wave trips, lane trips, measuring passes}, farther bands the same."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
emu = ctypes.CDLL(sys.argv[1])
pct = int(sys.argv[2]) if len(sys.argv) > 2 else 60
level = int(sys.argv[3]) if len(sys.argv) > 3 else 9
nblk = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dg = ctypes.CDLL(os.path.join(ROOT, "tools/libdatagen.so"))
dg.lz4amd_datagen.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, ctypes.c_uint32]
n = nblk * 262144
b = ctypes.create_string_buffer(n); dg.lz4amd_datagen(b, n, pct / 100.0, 0.0, 0)
datas = [b.raw[o:o + 262144] for o in range(0, n, 262144)]
caps = [len(d) + len(d) // 255 + 16 for d in datas]
srcs = [ctypes.create_string_buffer(d, len(d)) for d in datas]
dsts = [ctypes.create_string_buffer(c + 64) for c in caps]
k = len(datas)
sp = (ctypes.c_void_p * k)(*[ctypes.addressof(s) for s in srcs]); dp = (ctypes.c_void_p * k)(*[ctypes.addressof(d) for d in dsts])
ss = (ctypes.c_int32 * k)(*[len(d) for d in datas]); dc = (ctypes.c_int32 * k)(*caps); res = (ctypes.c_int32 * k)()
emu.emu_compress_hc_batch(sp, ss, dp, dc, res, k, 0, level)
st = (ctypes.c_ulonglong * 16).in_dll(emu, "lz4amd_emu_stats")
v = [x / k for x in st]
print("P%d level %d: %.0f bytes per block" % (pct, level, sum(res) / k))
print("band 0 : walks %.0f kept %.0f wave-trips %.0f lane-trips %.0f (%.1f lanes) measuring passes %.0f" % (v[0], v[1], v[2], v[3], v[3] / max(v[2], 1), v[4]))
print("first step: positions left to the walk loop %.0f" % v[10]); print("bands 1+: walks %.0f kept %.0f wave-trips %.0f lane-trips %.0f (%.1f lanes) measuring passes %.0f" % (v[5], v[6], v[7], v[8], v[8] / max(v[7], 1), v[9]))
