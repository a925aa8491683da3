"""Generate the committed golden vectors from the REAL reference (run in the build container).

    python tests/golden/make_golden.py

Needs oracle/_ref (built by `make -C oracle` from /root/reference).  Writes into tests/golden/:
  golden.json        sizes / md5 / xxh32 / known answers produced by the reference
  *.lz4blk           raw LZ4 blocks produced by the reference's LZ4_compress_default / _HC
  *.lz4              frames produced by the reference CLI (oracle/_ref/lz4)
The GPU box has no /root/reference: tests read only these files there.
"""
import ctypes
import hashlib
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")

ref = ctypes.CDLL(os.path.join(REF, "liblz4_ref.so"))
ref.LZ4_XXH32.restype = ctypes.c_uint32


def datagen(args):
    return subprocess.run([os.path.join(REF, "datagen")] + args, capture_output=True, check=True).stdout


def compress(data, hc_level=None):
    cap = ref.LZ4_compressBound(len(data))
    dst = ctypes.create_string_buffer(cap)
    if hc_level is None:
        r = ref.LZ4_compress_default(data, dst, len(data), cap)
    else:
        r = ref.LZ4_compress_HC(data, dst, len(data), cap, hc_level)
    return dst.raw[:r]


def md5(b):
    return hashlib.md5(b).hexdigest()


G = {"reference": "lz4/lz4 v1.10.0", "datagen": {}, "blocks": {}, "xxh32": {}, "frames": {}, "ratio": {}}

# --- datagen streams (inputs of every BASELINE config)
for args in (["-g65536", "-P50"], ["-g1000000", "-P60", "-s7"], ["-g4194304", "-P60"], ["-g300000", "-P20", "-s3"],
             ["-g262144", "-P60", "-s1"], ["-g40000", "-P90", "-s9"], ["-g1", "-P50"]):
    G["datagen"][" ".join(args)] = md5(datagen(args))

# --- blocks compressed by the reference: sizes + md5 (+ the bytes for the small ones)
cases = {
    "p50_64k": (["-g65536", "-P50"], None, True),
    "p60_256k_s1": (["-g262144", "-P60", "-s1"], None, False),
    "p60_4m": (["-g4194304", "-P60"], None, False),
    "p20_300k_s3": (["-g300000", "-P20", "-s3"], None, True),
    "p90_40k_s9": (["-g40000", "-P90", "-s9"], None, True),
    "p60_256k_s1_hc9": (["-g262144", "-P60", "-s1"], 9, True),
    "p50_64k_hc9": (["-g65536", "-P50"], 9, False),
}
for name, (args, lvl, keep) in cases.items():
    data = datagen(args)
    c = compress(data, lvl)
    G["blocks"][name] = {"datagen": " ".join(args), "level": lvl, "src_size": len(data), "src_md5": md5(data),
                         "csize": len(c), "c_md5": md5(c)}
    if keep:
        with open(os.path.join(HERE, name + ".lz4blk"), "wb") as f:
            f.write(c)
        G["blocks"][name]["file"] = name + ".lz4blk"

# --- known answers pinned by the reference's own tests
G["known"] = {
    "empty_block_hex": compress(b"").hex(),                       # tests/fuzzer.c:1125-1131
    "malformed_17_hex": bytes([0xEE] + [0] * 14 + [0x0E, 0x00]).hex(),  # fuzzer.c:1110-1119 must fail
}

# --- XXH32 (seed 0)
for s in (b"", b"a", b"abc", b"Nobody inspects the spammish repetition", bytes(range(16)), datagen(["-g100000", "-P50"])):
    key = s.hex() if len(s) <= 64 else "datagen -g100000 -P50"
    G["xxh32"][key] = ref.LZ4_XXH32(s, len(s), 0)

# --- frames written by the reference CLI
frame_cases = {
    "f_p60_600k_B4_BD_cs": (["-g600000", "-P60"], ["-B4", "-BD"]),          # 64 KB linked blocks + content checksum
    "f_p60_600k_B5_BI_BX": (["-g600000", "-P60"], ["-B5", "-BI", "-BX"]),   # 256 KB independent + block checksums
    "f_p50_100k_nocs": (["-g100000", "-P50"], ["-B4", "-BI", "--no-frame-crc"]),
}
for name, (dargs, largs) in frame_cases.items():
    data = datagen(dargs)
    out = subprocess.run([os.path.join(REF, "lz4"), "-c"] + largs, input=data, capture_output=True, check=True).stdout
    with open(os.path.join(HERE, name + ".lz4"), "wb") as f:
        f.write(out)
    G["frames"][name] = {"datagen": " ".join(dargs), "lz4_args": " ".join(largs), "src_size": len(data),
                         "src_md5": md5(data), "frame_size": len(out), "header_hex": out[:7].hex(),
                         "trailer_hex": out[-8:].hex()}

# --- reference ratios on the benchmark inputs (what the +-3 % window is measured against)
def ratio(dargs, bs, lvl=None):
    data = datagen(dargs)
    tot = sum(len(compress(data[o:o + bs], lvl)) for o in range(0, len(data), bs))
    return {"src": len(data), "block": bs, "level": lvl, "csize": tot}
G["ratio"]["p60_16m_4m_blocks"] = ratio(["-g16M", "-P60"], 4 << 20)
G["ratio"]["p50_4m_64k_blocks"] = ratio(["-g4M", "-P50"], 64 << 10)
G["ratio"]["p60_4m_256k_blocks_hc9"] = ratio(["-g4M", "-P60"], 256 << 10, 9)
# LZ4_compress_HC at other depths and on other inputs (tests/test_gpu_hc.py, test_kernels_emulated.py)
G["ratio"]["p60_4m_256k_blocks_hc3"] = ratio(["-g4M", "-P60"], 256 << 10, 3)
G["ratio"]["p60_4m_256k_blocks_hc6"] = ratio(["-g4M", "-P60"], 256 << 10, 6)
G["ratio"]["p90_4m_256k_blocks_hc9"] = ratio(["-g4M", "-P90"], 256 << 10, 9)
G["ratio"]["p20_2m_256k_blocks_hc9"] = ratio(["-g2M", "-P20"], 256 << 10, 9)
G["ratio"]["p60_8m_4m_blocks_hc9"] = ratio(["-g8M", "-P60"], 4 << 20, 9)
G["ratio"]["p50_1m_64k_blocks_hc9"] = ratio(["-g1M", "-P50"], 64 << 10, 9)
# level 2 = LZ4MID (lz4hc.c:93-95, 472-773), and the fast codec on the same blocks
for pct, size in ((60, "4M"), (90, "4M"), (20, "2M")):
    G["ratio"]["p%d_%s_256k_blocks_hc2" % (pct, size.lower())] = ratio(["-g" + size, "-P%d" % pct], 256 << 10, 2)
    G["ratio"]["p%d_%s_256k_blocks_fast" % (pct, size.lower())] = ratio(["-g" + size, "-P%d" % pct], 256 << 10)

with open(os.path.join(HERE, "golden.json"), "w") as f:
    json.dump(G, f, indent=1, sort_keys=True)
print("wrote", os.path.join(HERE, "golden.json"))
