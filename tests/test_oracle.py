"""The oracle (oracle/lz4_oracle.c) pinned against the reference's golden vectors and, when
oracle/_ref is available, against the real reference live."""
import ctypes
import os
import random

import pytest

from conftest import GOLDEN_DIR, md5

DATAGEN_ARGS = {"-g": None}


def _parse_dg(s):
    n, p, seed = 0, 50, 0
    for tok in s.split():
        if tok.startswith("-g"):
            v = tok[2:]
            mult = {"K": 1 << 10, "M": 1 << 20}.get(v[-1], 1)
            n = int(v[:-1]) * mult if v[-1] in "KM" else int(v)
        elif tok.startswith("-P"):
            p = int(tok[2:])
        elif tok.startswith("-s"):
            seed = int(tok[2:])
    return n, p, seed


def test_datagen_matches_reference_md5(golden, datagen):
    for args, want in golden["datagen"].items():
        n, p, seed = _parse_dg(args)
        assert md5(datagen(n, p, seed)) == want, args


def test_oracle_compress_is_byte_identical_to_reference_golden(golden, datagen, ocodec):
    for name, g in golden["blocks"].items():
        if g["level"] is not None:
            continue
        n, p, seed = _parse_dg(g["datagen"])
        data = datagen(n, p, seed)
        assert md5(data) == g["src_md5"]
        r, c = ocodec.compress(data)
        assert r == g["csize"], name
        assert md5(c) == g["c_md5"], name


def test_oracle_decodes_reference_blocks(golden, datagen, ocodec):
    for name, g in golden["blocks"].items():
        if "file" not in g:
            continue
        comp = open(os.path.join(GOLDEN_DIR, g["file"]), "rb").read()
        assert len(comp) == g["csize"] and md5(comp) == g["c_md5"]
        r, out = ocodec.decompress(comp, g["src_size"])
        assert r == g["src_size"] and md5(out) == g["src_md5"], name


def test_known_answers(golden, ocodec):
    r, c = ocodec.compress(b"")
    assert r == 1 and c.hex() == golden["known"]["empty_block_hex"]          # fuzzer.c:1125-1131
    bad = bytes.fromhex(golden["known"]["malformed_17_hex"])
    assert ocodec.decompress(bad, 100)[0] < 0                                  # fuzzer.c:1110-1119
    assert ocodec.decompress(b"\x00", 0)[0] == 0                               # lz4.c:2064-2068
    assert ocodec.decompress(b"\x01", 0)[0] < 0


def test_xxh32_golden(golden, oracle, datagen):
    for key, want in golden["xxh32"].items():
        data = datagen(100000, 50) if key.startswith("datagen") else bytes.fromhex(key)
        assert oracle.lz4o_xxh32(data, len(data), 0) == want


def test_frame_oracle_decodes_reference_cli_frames(golden, oracle, datagen):
    for name, g in golden["frames"].items():
        frame = open(os.path.join(GOLDEN_DIR, name + ".lz4"), "rb").read()
        assert frame[:7].hex() == g["header_hex"]
        out = ctypes.create_string_buffer(g["src_size"] + 16)
        used = ctypes.c_size_t()
        r = oracle.lz4o_frame_decompress(out, g["src_size"], frame, len(frame), ctypes.byref(used))
        assert r == g["src_size"] and used.value == len(frame), name
        assert md5(out.raw[:r]) == g["src_md5"]
        # a flipped payload byte must be caught by the content / block checksum
        if "no-frame-crc" not in g["lz4_args"]:
            bad = bytearray(frame); bad[len(bad) // 2] ^= 0x55
            assert oracle.lz4o_frame_decompress(out, g["src_size"], bytes(bad), len(bad), None) == ctypes.c_size_t(-1).value


def test_frame_header_checksum_known_values(oracle):
    # HC byte = (XXH32(FLG|BD) >> 8) & 0xFF
    for flg, bd, hc in ((0x44, 0x70, 0x1D), (0x64, 0x70, 0xB9), (0x60, 0x70, 0x73), (0x64, 0x40, 0xA7)):
        assert (oracle.lz4o_xxh32(bytes([flg, bd]), 2, 0) >> 8) & 0xFF == hc


# ------------------------------------------------------------------ live against oracle/_ref
def test_live_compress_byte_identical(reflib, ocodec, datagen):
    rnd = random.Random(1)
    cases = [datagen(n, p, n & 7) for n in (0, 1, 12, 13, 14, 100, 4096, 65535, 65546, 65547, 70000, 300000) for p in (0, 20, 50, 60, 90)]
    cases += [b"\x00" * 100000, b"abcd" * 30000, os.urandom(70000), b"a" * 70000 + os.urandom(10) + b"a" * 70000]
    for data in cases:
        n = len(data)
        cap = reflib.LZ4_compressBound(n)
        for accel in (1, 4):
            d = ctypes.create_string_buffer(max(cap, 1))
            r = reflib.LZ4_compress_fast(data, d, n, cap, accel)
            ro, co = ocodec.compress(data, accel=accel)
            assert r == ro and d.raw[:r] == co
        for lim in (r, r - 1, r // 2, 1):
            if lim < 0:
                continue
            d = ctypes.create_string_buffer(max(lim, 1))
            a = reflib.LZ4_compress_fast(data, d, n, lim, 1)
            b, cb = ocodec.compress(data, cap=lim)
            assert a == b and d.raw[:max(a, 0)] == cb
        _ = rnd


def test_live_decoder_agrees_on_hostile_input(reflib, ocodec, datagen):
    rnd = random.Random(7)
    data = datagen(200000, 60, 1)
    _, comp = ocodec.compress(data)
    for t in range(1500):
        cc = bytearray(comp[:rnd.randint(1, len(comp))] if t % 2 else comp)
        for _ in range(rnd.randint(1, 4)):
            cc[rnd.randrange(len(cc))] = rnd.randrange(256)
        cc = bytes(cc)
        d = ctypes.create_string_buffer(200000 + 8)
        a = reflib.LZ4_decompress_safe(cc, d, len(cc), 200000)
        b, out = ocodec.decompress(cc, 200000)
        # the oracle applies the safe-loop rules everywhere: it may reject a stream the reference's
        # fast loop lets through, never the other way round, and agrees on everything it accepts
        if b >= 0:
            assert a == b and d.raw[:a] == out
        if a < 0:
            assert b < 0


def test_live_frame_roundtrip_both_directions(reflib, oracle, datagen):
    data = datagen(700000, 60, 3)
    for bsid, bchk, cchk, csz in ((4, 0, 1, 0), (5, 1, 1, 1), (7, 0, 0, 0), (6, 1, 0, 1)):
        cap = oracle.lz4o_frame_bound(len(data), bsid, bchk, cchk)
        buf = ctypes.create_string_buffer(cap)
        n = oracle.lz4o_frame_compress(buf, cap, data, len(data), bsid, bchk, cchk, csz)
        assert n > 0
        # reference decodes the oracle's frame
        dctx = ctypes.c_void_p()
        reflib.LZ4F_createDecompressionContext.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        reflib.LZ4F_decompress.restype = ctypes.c_size_t
        reflib.LZ4F_decompress.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
        assert reflib.LZ4F_createDecompressionContext(ctypes.byref(dctx), 100) == 0
        out = ctypes.create_string_buffer(len(data) + 16)
        dsz, ssz = ctypes.c_size_t(len(data) + 16), ctypes.c_size_t(n)
        rc = reflib.LZ4F_decompress(dctx, out, ctypes.byref(dsz), buf.raw[:n], ctypes.byref(ssz), None)
        assert rc == 0 and dsz.value == len(data) and out.raw[:dsz.value] == data
        reflib.LZ4F_freeDecompressionContext.argtypes = [ctypes.c_void_p]
        reflib.LZ4F_freeDecompressionContext(dctx)
