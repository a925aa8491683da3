"""The host-side entry points of the ABI's long tail (lz4_amd/csrc/lz4_compat_api.c) that need no device:
LZ4_decompress_safe_partial*, the deprecated LZ4_decompress_fast* family and the LZ4_XXH32 / LZ4_XXH64 exports,
compared call by call with the real reference (oracle/_ref) on reference-compressed blocks."""
import ctypes
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vp, ci = ctypes.c_void_p, ctypes.c_int


@pytest.fixture(scope="module")
def ours():
    so = os.path.join(ROOT, "lz4_amd", "liblz4_amd.so")
    if not os.path.exists(so):
        pytest.skip("liblz4_amd.so not built")
    L = ctypes.CDLL(so)
    L.LZ4_XXH32.restype = ctypes.c_uint32
    L.LZ4_XXH32.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32]
    L.LZ4_XXH64.restype = ctypes.c_uint64
    L.LZ4_XXH64.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64]
    return L


def _blocks(reflib, datagen):
    rnd = random.Random(3)
    datas = [datagen(n, p, s) for n, p, s in ((70000, 50, 1), (3000, 80, 2), (200, 30, 3), (65536, 95, 4), (40000, 0, 5))]
    datas += [b"a" * 5000, b"abcdefgh" * 900 + bytes(rnd.getrandbits(8) for _ in range(700)), b"x" * 13]
    out = []
    for d in datas:
        cap = reflib.LZ4_compressBound(len(d))
        cb = ctypes.create_string_buffer(cap)
        n = reflib.LZ4_compress_default(d, cb, len(d), cap)
        assert n > 0
        out.append((d, cb.raw[:n]))
    return out


def test_partial_decode_matches_the_reference(ours, reflib, datagen):
    for d, c in _blocks(reflib, datagen):
        for target in sorted({0, 1, 5, len(d) // 3, len(d) - 1, len(d), len(d) + 10}):
            for cut in (len(c), max(1, len(c) * 2 // 3)):            # whole block, truncated input
                a = ctypes.create_string_buffer(len(d) + 64); b = ctypes.create_string_buffer(len(d) + 64)
                ra = ours.LZ4_decompress_safe_partial(c[:cut], a, cut, target, len(d) + 10)
                rb = reflib.LZ4_decompress_safe_partial(c[:cut], b, cut, target, len(d) + 10)
                if rb >= 0:
                    assert ra == rb, (len(d), target, cut, ra, rb)
                    assert a.raw[:ra] == b.raw[:rb] == d[:ra]
                    assert a.raw[ra:ra + 16] == b"\x00" * 16          # nothing written behind what was asked for
                else:
                    assert ra < 0


def test_partial_decode_with_dictionary(ours, reflib, datagen):
    data = datagen(90000, 60, 8)
    dict_, body = data[:30000], data[30000:]
    s = reflib.LZ4_createStream
    reflib.LZ4_createStream.restype = vp
    st = reflib.LZ4_createStream()
    reflib.LZ4_loadDict.argtypes = [vp, ctypes.c_char_p, ci]
    reflib.LZ4_compress_fast_continue.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, ci, ci, ci]
    dbuf = ctypes.create_string_buffer(dict_, len(dict_))
    reflib.LZ4_loadDict(st, dbuf, len(dict_))
    cap = reflib.LZ4_compressBound(len(body)); cb = ctypes.create_string_buffer(cap)
    n = reflib.LZ4_compress_fast_continue(st, body, cb, len(body), cap, 1)
    assert n > 0
    for target in (100, 20000, len(body)):
        a = ctypes.create_string_buffer(len(body) + 16)
        r = ours.LZ4_decompress_safe_partial_usingDict(cb.raw[:n], a, n, target, len(body), dbuf, len(dict_))
        assert r == target and a.raw[:r] == body[:r]


def test_deprecated_fast_decoders(ours, reflib, datagen):
    for d, c in _blocks(reflib, datagen):
        if len(d) < 14:
            continue
        a = ctypes.create_string_buffer(len(d) + 16)
        assert ours.LZ4_decompress_fast(c, a, len(d)) == len(c)      # returns the input bytes read
        assert a.raw[:len(d)] == d and a.raw[len(d)] == 0
        a = ctypes.create_string_buffer(len(d) + 16)
        assert ours.LZ4_decompress_fast(c, a, len(d) - 1) < 0        # fuzzer.c:532-542
        assert a.raw[len(d) - 1] == 0
        assert ours.LZ4_decompress_fast(c, a, len(d) + 1) < 0
        assert ours.LZ4_uncompress(c, a, len(d)) == len(c)


def test_xxh_exports(ours, reflib):
    reflib.LZ4_XXH32.restype = ctypes.c_uint32
    reflib.LZ4_XXH32.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32]
    reflib.LZ4_XXH64.restype = ctypes.c_uint64
    reflib.LZ4_XXH64.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64]
    rnd = random.Random(1)
    for n in (0, 1, 3, 4, 15, 16, 17, 31, 32, 33, 63, 64, 100, 1000, 65537):
        b = bytes(rnd.getrandbits(8) for _ in range(n))
        for seed in (0, 1, 0x9E3779B1):
            assert ours.LZ4_XXH32(b, n, seed) == reflib.LZ4_XXH32(b, n, seed)
            assert ours.LZ4_XXH64(b, n, seed) == reflib.LZ4_XXH64(b, n, seed)
    # streaming states
    ours.LZ4_XXH64_createState.restype = vp
    st = ours.LZ4_XXH64_createState()
    ours.LZ4_XXH64_reset.argtypes = [vp, ctypes.c_uint64]
    ours.LZ4_XXH64_update.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t]
    ours.LZ4_XXH64_digest.argtypes = [vp]; ours.LZ4_XXH64_digest.restype = ctypes.c_uint64
    ours.LZ4_XXH64_freeState.argtypes = [vp]
    b = bytes(rnd.getrandbits(8) for _ in range(5000))
    ours.LZ4_XXH64_reset(st, 7)
    for i in range(0, 5000, 37):
        ours.LZ4_XXH64_update(st, b[i:i + 37], len(b[i:i + 37]))
    assert ours.LZ4_XXH64_digest(st) == reflib.LZ4_XXH64(b, 5000, 7)
    ours.LZ4_XXH64_freeState(st)


def test_lz4file_argument_and_io_errors(ours, tmp_path):
    """include/lz4file.h (lib/lz4file.c:73-138, 217-279): NULL arguments -> parameter_null, a file shorter than an empty frame
    -> io_read, a file that does not start with a frame -> the header's error; no state is left behind (*out == NULL).  These
    paths end before any block is coded: no device needed."""
    libc = ctypes.CDLL(None)
    libc.fopen.restype = vp
    libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    libc.fclose.argtypes = [vp]
    for f in (ours.LZ4F_readOpen, ours.LZ4F_writeOpen, ours.LZ4F_readClose, ours.LZ4F_writeClose, ours.LZ4F_read, ours.LZ4F_write):
        f.restype = ctypes.c_size_t
    ours.LZ4F_readOpen.argtypes = [ctypes.POINTER(vp), vp]
    ours.LZ4F_writeOpen.argtypes = [ctypes.POINTER(vp), vp, vp]
    ours.LZ4F_readClose.argtypes = [vp]
    ours.LZ4F_writeClose.argtypes = [vp]
    ours.LZ4F_read.argtypes = [vp, vp, ctypes.c_size_t]
    ours.LZ4F_write.argtypes = [vp, vp, ctypes.c_size_t]
    ours.LZ4F_getErrorName.restype = ctypes.c_char_p
    ours.LZ4F_getErrorName.argtypes = [ctypes.c_size_t]
    ours.LZ4F_isError.argtypes = [ctypes.c_size_t]
    name = lambda code: ours.LZ4F_getErrorName(code)
    h = vp()
    assert name(ours.LZ4F_readOpen(ctypes.byref(h), None)) == b"ERROR_parameter_null"
    assert name(ours.LZ4F_readOpen(None, None)) == b"ERROR_parameter_null"
    assert name(ours.LZ4F_writeOpen(ctypes.byref(h), None, None)) == b"ERROR_parameter_null"
    assert name(ours.LZ4F_readClose(None)) == b"ERROR_parameter_null" and name(ours.LZ4F_writeClose(None)) == b"ERROR_parameter_null"
    assert name(ours.LZ4F_read(None, None, 0)) == b"ERROR_parameter_null" and name(ours.LZ4F_write(None, None, 0)) == b"ERROR_parameter_null"
    short = tmp_path / "short.lz4"
    short.write_bytes(b"\x04\x22\x4d\x18\x60")
    fp = libc.fopen(str(short).encode(), b"rb")
    h = vp(1)
    assert name(ours.LZ4F_readOpen(ctypes.byref(h), fp)) == b"ERROR_io_read" and not h.value
    libc.fclose(fp)
    junk = tmp_path / "junk.lz4"
    junk.write_bytes(b"not an lz4 frame at all, twenty+ bytes")
    fp = libc.fopen(str(junk).encode(), b"rb")
    h = vp(1)
    rc = ours.LZ4F_readOpen(ctypes.byref(h), fp)
    assert ours.LZ4F_isError(rc) and not h.value
    libc.fclose(fp)


def test_fast_continue_with_a_long_prefix_and_a_buffer_switch(ours, reflib, datagen):
    """LZ4_decompress_fast_continue (lz4.c:2798-2830): a first block of more than 64 KB, then a block decoded into ANOTHER buffer
    that copies from the tail of the first - the dictionary is the LAST 64 KB of the previous output, not its first."""
    vp_ = ctypes.c_void_p
    data = datagen(140000, 70, 12)
    a, b = data[:100000], data[100000:]
    reflib.LZ4_createStream.restype = vp_
    reflib.LZ4_compress_fast_continue.argtypes = [vp_, ctypes.c_char_p, ctypes.c_char_p, ci, ci, ci]
    reflib.LZ4_freeStream.argtypes = [vp_]
    s = reflib.LZ4_createStream()
    srcbuf = ctypes.create_string_buffer(data, len(data))                 # both blocks contiguous: the second sees the first as its prefix
    cap = reflib.LZ4_compressBound(len(a))
    ca, cb = ctypes.create_string_buffer(cap), ctypes.create_string_buffer(cap)
    base = ctypes.addressof(srcbuf)
    na = reflib.LZ4_compress_fast_continue(s, ctypes.cast(base, ctypes.c_char_p), ca, len(a), cap, 1)
    nb = reflib.LZ4_compress_fast_continue(s, ctypes.cast(base + len(a), ctypes.c_char_p), cb, len(b), cap, 1)
    reflib.LZ4_freeStream(s)
    assert na > 0 and nb > 0
    for lib in (reflib, ours):
        lib.LZ4_createStreamDecode.restype = vp_
        lib.LZ4_decompress_fast_continue.argtypes = [vp_, ctypes.c_char_p, ctypes.c_char_p, ci]
        lib.LZ4_freeStreamDecode.argtypes = [vp_]
        sd = lib.LZ4_createStreamDecode()
        o1, o2 = ctypes.create_string_buffer(len(a) + 16), ctypes.create_string_buffer(len(b) + 16)      # two separate buffers
        assert lib.LZ4_decompress_fast_continue(sd, ca.raw[:na], o1, len(a)) == na
        assert lib.LZ4_decompress_fast_continue(sd, cb.raw[:nb], o2, len(b)) == nb
        assert o1.raw[:len(a)] == a and o2.raw[:len(b)] == b, lib
        lib.LZ4_freeStreamDecode(sd)


def test_lz4file_read_open_returns_ok(ours, tmp_path):
    """LZ4F_readOpen (lib/lz4file.c:73-140) returns LZ4F_OK_NoError on success, not LZ4F_getFrameInfo's size hint: callers may
    compare with 0.  Only the header is touched here: no device needed."""
    libc = ctypes.CDLL(None)
    libc.fopen.restype = vp; libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]; libc.fclose.argtypes = [vp]
    ours.LZ4F_readOpen.restype = ctypes.c_size_t; ours.LZ4F_readOpen.argtypes = [ctypes.POINTER(vp), vp]
    ours.LZ4F_readClose.restype = ctypes.c_size_t; ours.LZ4F_readClose.argtypes = [vp]
    src = os.path.join(ROOT, "tests", "golden", "f_p60_600k_B4_BD_cs.lz4")
    fp = libc.fopen(src.encode(), b"rb")
    h = vp()
    assert ours.LZ4F_readOpen(ctypes.byref(h), fp) == 0 and h.value
    assert ours.LZ4F_readClose(h) == 0
    libc.fclose(fp)
