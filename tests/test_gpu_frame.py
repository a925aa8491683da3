"""Frame container and XXH32 on the GPU path, through the C ABI of
include/lz4frame.h and include/lz4amd.h.  Parity directions as in SURVEY 8c: our frames decode with
the oracle's (== reference's) frame decoder, frames written by the reference CLI decode here."""
import ctypes
import os
import random

import pytest

pytestmark = pytest.mark.gpu


class FrameInfo(ctypes.Structure):
    _fields_ = [("blockSizeID", ctypes.c_int), ("blockMode", ctypes.c_int), ("contentChecksumFlag", ctypes.c_int),
                ("frameType", ctypes.c_int), ("contentSize", ctypes.c_ulonglong), ("dictID", ctypes.c_uint),
                ("blockChecksumFlag", ctypes.c_int)]


class Prefs(ctypes.Structure):
    _fields_ = [("frameInfo", FrameInfo), ("compressionLevel", ctypes.c_int), ("autoFlush", ctypes.c_uint),
                ("favorDecSpeed", ctypes.c_uint), ("reserved", ctypes.c_uint * 3)]


@pytest.fixture(scope="module")
def L():
    import lz4_amd
    lib = lz4_amd.lib()
    st, vp = ctypes.c_size_t, ctypes.c_void_p
    lib.LZ4F_compressFrameBound.restype = st
    lib.LZ4F_compressFrameBound.argtypes = [st, ctypes.POINTER(Prefs)]
    lib.LZ4F_compressFrame.restype = st
    lib.LZ4F_compressFrame.argtypes = [ctypes.c_char_p, st, ctypes.c_char_p, st, ctypes.POINTER(Prefs)]
    lib.LZ4F_isError.argtypes = [st]
    lib.LZ4F_getErrorName.restype = ctypes.c_char_p
    lib.LZ4F_getErrorName.argtypes = [st]
    lib.LZ4F_createDecompressionContext.restype = st
    lib.LZ4F_createDecompressionContext.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
    lib.LZ4F_freeDecompressionContext.argtypes = [vp]
    lib.LZ4F_decompress.restype = st
    lib.LZ4F_decompress.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(st), ctypes.c_char_p, ctypes.POINTER(st), vp]
    lib.LZ4F_getFrameInfo.restype = st
    lib.LZ4F_getFrameInfo.argtypes = [vp, ctypes.POINTER(FrameInfo), ctypes.c_char_p, ctypes.POINTER(st)]
    return lib


def compress_frame(L, data, level=0, **kw):
    p = Prefs()
    p.compressionLevel = level
    for k, v in kw.items():
        setattr(p.frameInfo, k, v)
    cap = L.LZ4F_compressFrameBound(len(data), ctypes.byref(p))
    assert not L.LZ4F_isError(cap)
    dst = ctypes.create_string_buffer(cap + 16)
    n = L.LZ4F_compressFrame(dst, cap, data, len(data), ctypes.byref(p))
    assert not L.LZ4F_isError(n), L.LZ4F_getErrorName(n)
    assert dst.raw[cap:cap + 16] == b"\0" * 16
    return dst.raw[:n]


def decompress_frame(L, frame, total, chunk_rng=None, expect_error=False):
    d = ctypes.c_void_p()
    assert L.LZ4F_createDecompressionContext(ctypes.byref(d), 100) == 0
    out = bytearray()
    pos = 0
    try:
        for _ in range(100000):
            take = len(frame) - pos if chunk_rng is None else min(len(frame) - pos, chunk_rng.randint(1, 70000))
            src = frame[pos:pos + take]
            dst = ctypes.create_string_buffer(max(1, total if chunk_rng is None else chunk_rng.randint(1, 200000)))
            dsz, ssz = ctypes.c_size_t(len(dst)), ctypes.c_size_t(len(src))
            r = L.LZ4F_decompress(d, dst, ctypes.byref(dsz), src, ctypes.byref(ssz), None)
            if L.LZ4F_isError(r):
                assert expect_error, L.LZ4F_getErrorName(r)
                return None
            out += dst.raw[:dsz.value]
            pos += ssz.value
            if r == 0:
                break
            assert ssz.value or dsz.value or pos < len(frame), "no progress"
        assert not expect_error
        return bytes(out), pos
    finally:
        L.LZ4F_freeDecompressionContext(d)


def test_xxh32_batch_on_gpu(oracle, datagen):
    import torch
    import lz4_amd
    ctx = lz4_amd.Context(0)
    datas = [b"", b"a", b"abc", bytes(range(16)), datagen(1000, 50, 1), datagen(1024, 50, 2), datagen(1025, 50, 3),
             datagen(4 << 20, 60, 4), datagen(65536, 50, 5)[3:], b"z" * 15]
    bufs = [torch.frombuffer(bytearray(d + b"\0"), dtype=torch.uint8).cuda() for d in datas]
    t = lz4_amd.BlockTable([b.data_ptr() for b in bufs], [len(d) for d in datas], [0] * len(datas), [0] * len(datas))
    plan = lz4_amd.Plan(ctx, lz4_amd.OP_XXH32, t)
    plan.launch(torch.cuda.current_stream().cuda_stream)
    res = plan.results(torch.cuda.current_stream().cuda_stream)
    for d, r in zip(datas, res):
        assert (r & 0xFFFFFFFF) == oracle.lz4o_xxh32(d, len(d), 0)


@pytest.mark.parametrize("kw", [
    dict(), dict(blockSizeID=7, contentChecksumFlag=1), dict(blockSizeID=5, blockChecksumFlag=1),
    dict(blockSizeID=4, contentChecksumFlag=1, blockChecksumFlag=1, contentSize=1), dict(blockSizeID=6, blockMode=1)])
def test_our_frames_decode_with_the_oracle_and_here(L, oracle, datagen, kw):
    for n, pct in ((0, 50), (1, 50), (100, 50), (65536, 50), (65537, 60), (700000, 60), (9 << 20, 60)):
        data = datagen(n, pct, n % 7)
        frame = compress_frame(L, data, **kw)
        assert frame[:4] == bytes.fromhex("04224d18")
        bs = {0: 65536, 4: 65536, 5: 262144, 6: 1 << 20, 7: 4 << 20}[kw.get("blockSizeID", 0)]
        one_block = n <= 65536 or n <= bs             # lz4frame.c:388-398 shrinks the block size id to fit a small input
        assert bool(frame[4] & 0x20) == (kw.get("blockMode", 0) == 1 or one_block)     # linked unless asked / single block
        out = ctypes.create_string_buffer(n + 1)
        used = ctypes.c_size_t()
        r = oracle.lz4o_frame_decompress(out, n, frame, len(frame), ctypes.byref(used))
        assert r == n and used.value == len(frame) and out.raw[:n] == data
        got, pos = decompress_frame(L, frame, n)
        assert got == data and pos == len(frame)


def test_hc_levels_in_frames(L, oracle, datagen):
    """compressionLevel >= LZ4HC_CLEVEL_MIN selects the HC compressor (lz4frame.c:943-958): the frame is smaller
    and still decodes with the oracle's frame decoder and here."""
    data = datagen(3 << 20, 60, 5)
    for kw in (dict(blockSizeID=5, blockMode=1, contentChecksumFlag=1), dict(blockSizeID=6)):
        fast = compress_frame(L, data, **kw)
        hc = compress_frame(L, data, level=9, **kw)
        assert len(hc) < 0.85 * len(fast)
        assert hc[:7] == fast[:7]                                      # same header: the level is not recorded
        out = ctypes.create_string_buffer(len(data) + 1)
        used = ctypes.c_size_t()
        r = oracle.lz4o_frame_decompress(out, len(data), hc, len(hc), ctypes.byref(used))
        assert r == len(data) and used.value == len(hc) and out.raw[:len(data)] == data
        got, pos = decompress_frame(L, hc, len(data))
        assert got == data and pos == len(hc)
    # linked HC blocks see the 64 KB before them: smaller than the same blocks compressed independently
    assert len(compress_frame(L, data, level=9, blockSizeID=4)) < len(compress_frame(L, data, level=9, blockSizeID=4, blockMode=1))


def test_linked_frame_with_stored_blocks_in_the_chain(L, oracle, datagen):
    """Linked blocks decode in one launch (lz4amd_plan_create_decompress_chained); incompressible blocks are stored
    (lz4frame.c:896-899) and sit in the middle of the chain: the blocks behind them copy from their bytes."""
    rng = random.Random(17)
    data = datagen(200000, 60, 1) + rng.randbytes(150000) + datagen(300000, 60, 2) + rng.randbytes(70000) + datagen(100000, 50, 3)
    for kw in (dict(blockSizeID=4, contentChecksumFlag=1), dict(blockSizeID=5, blockChecksumFlag=1)):
        frame = compress_frame(L, data, **kw)
        assert not frame[4] & 0x20                                            # linked
        out = ctypes.create_string_buffer(len(data) + 1)
        used = ctypes.c_size_t()
        r = oracle.lz4o_frame_decompress(out, len(data), frame, len(frame), ctypes.byref(used))
        assert r == len(data) and out.raw[:len(data)] == data
        got, pos = decompress_frame(L, frame, len(data))
        assert got == data and pos == len(frame)
        got, pos = decompress_frame(L, frame, len(data), chunk_rng=rng)        # batches that start anywhere in the chain
        assert got == data and pos == len(frame)


def test_frame_header_known_answers(L, datagen):
    # FLG 0x64 (v1, independent, content checksum) BD 0x70 (4 MB) -> HC 0xB9; FLG 0x60 -> 0x73
    data = datagen(5 << 20, 60, 0)
    assert compress_frame(L, data, blockSizeID=7, contentChecksumFlag=1, blockMode=1)[:7] == bytes.fromhex("04224d186470b9")
    assert compress_frame(L, data, blockSizeID=7, blockMode=1)[:7] == bytes.fromhex("04224d18607073")
    # BASELINE configs[2]: 4 MB linked blocks + content checksum -> 04 22 4D 18 44 70 1D
    assert compress_frame(L, data, blockSizeID=7, contentChecksumFlag=1)[:7] == bytes.fromhex("04224d1844701d")


def test_reference_cli_frames_decode_here(L, golden, datagen):
    from conftest import GOLDEN_DIR
    rng = random.Random(3)
    for name, g in golden["frames"].items():
        frame = open(os.path.join(GOLDEN_DIR, name + ".lz4"), "rb").read()
        args = g["datagen"].split()
        data = datagen(int(args[0][2:]), int(args[1][2:]), 0)
        got, pos = decompress_frame(L, frame, len(data))
        assert got == data and pos == len(frame), name
        got, pos = decompress_frame(L, frame, len(data), chunk_rng=rng)       # arbitrary input / output chunking
        assert got == data and pos == len(frame), name


def test_frame_errors(L, datagen):
    data = datagen(300000, 60, 1)
    frame = bytearray(compress_frame(L, data, blockSizeID=4, contentChecksumFlag=1, blockChecksumFlag=1))
    bad = bytearray(frame); bad[6] ^= 1                       # header checksum
    assert decompress_frame(L, bytes(bad), len(data), expect_error=True) is None
    bad = bytearray(frame); bad[-1] ^= 1                      # content checksum
    assert decompress_frame(L, bytes(bad), len(data), expect_error=True) is None
    bad = bytearray(frame); bad[40] ^= 0x55                   # block payload -> block checksum
    assert decompress_frame(L, bytes(bad), len(data), expect_error=True) is None
    bad = bytearray(frame); bad[0] ^= 1                       # magic
    assert decompress_frame(L, bytes(bad), len(data), expect_error=True) is None
    # two frames back to back: the decoder stops exactly at the end of the first
    got, pos = decompress_frame(L, bytes(frame) + bytes(frame), len(data))
    assert got == data and pos == len(frame)


def test_decompress_safe_using_dict(L, golden):
    """LZ4_decompress_safe_usingDict (lz4.c:2719): linked blocks of a reference-written frame, each
    decoded through the classic host-pointer API with the previous output as dictionary."""
    from conftest import GOLDEN_DIR
    from test_kernels_emulated import _frame_blocks
    import hashlib
    frame = open(os.path.join(GOLDEN_DIR, "f_p60_600k_B4_BD_cs.lz4"), "rb").read()
    indep, blocks = _frame_blocks(frame)
    L.LZ4_decompress_safe_usingDict.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    out = b""
    for raw, payload in blocks[:4]:
        dst = ctypes.create_string_buffer(65536)
        r = L.LZ4_decompress_safe_usingDict(payload, dst, len(payload), 65536, out[-65536:] if out else None, min(len(out), 65536))
        assert r > 0
        out += dst.raw[:r]
    # the frame's content is datagen -g600000 -P60: compare the prefix through the golden md5 of the whole
    full, _ = decompress_frame(L, frame, 600000)
    assert out == full[:len(out)] and hashlib.md5(full).hexdigest() == golden["frames"]["f_p60_600k_B4_BD_cs"]["src_md5"]


class COpts(ctypes.Structure):
    _fields_ = [("stableSrc", ctypes.c_uint), ("reserved", ctypes.c_uint * 3)]


def stream_compress(L, data, rng, level=0, auto_flush=0, flush_every=0, **kw):
    """LZ4F_compressBegin / Update (arbitrary input pieces) / flush / End, the way lz4io.c and frametest.c drive it."""
    st, vp = ctypes.c_size_t, ctypes.c_void_p
    L.LZ4F_createCompressionContext.restype = st
    L.LZ4F_createCompressionContext.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
    L.LZ4F_freeCompressionContext.argtypes = [vp]
    for name in ("LZ4F_compressBegin",):
        getattr(L, name).restype = st
        getattr(L, name).argtypes = [vp, ctypes.c_char_p, st, ctypes.POINTER(Prefs)]
    L.LZ4F_compressBound.restype = st
    L.LZ4F_compressBound.argtypes = [st, ctypes.POINTER(Prefs)]
    L.LZ4F_compressUpdate.restype = st
    L.LZ4F_compressUpdate.argtypes = [vp, ctypes.c_char_p, st, ctypes.c_char_p, st, vp]
    for name in ("LZ4F_flush", "LZ4F_compressEnd"):
        getattr(L, name).restype = st
        getattr(L, name).argtypes = [vp, ctypes.c_char_p, st, vp]
    p = Prefs()
    p.compressionLevel = level
    p.autoFlush = auto_flush
    for k, v in kw.items():
        setattr(p.frameInfo, k, v)
    c = vp()
    assert L.LZ4F_createCompressionContext(ctypes.byref(c), 100) == 0
    out = bytearray()
    hdr = ctypes.create_string_buffer(32)
    n = L.LZ4F_compressBegin(c, hdr, 32, ctypes.byref(p))
    assert not L.LZ4F_isError(n), L.LZ4F_getErrorName(n)
    out += hdr.raw[:n]
    pos, calls = 0, 0
    while pos < len(data):
        take = min(len(data) - pos, rng.choice((1, 7, 1000, 65536, 65537, 300000, 1 << 20)))
        cap = L.LZ4F_compressBound(take, ctypes.byref(p))
        dst = ctypes.create_string_buffer(cap + 8)
        n = L.LZ4F_compressUpdate(c, dst, cap, data[pos:pos + take], take, None)
        assert not L.LZ4F_isError(n), L.LZ4F_getErrorName(n)
        assert dst.raw[cap:] == b"\0" * 8
        out += dst.raw[:n]
        pos += take
        calls += 1
        if flush_every and calls % flush_every == 0:
            cap = L.LZ4F_compressBound(0, ctypes.byref(p))
            dst = ctypes.create_string_buffer(cap)
            n = L.LZ4F_flush(c, dst, cap, None)
            assert not L.LZ4F_isError(n)
            out += dst.raw[:n]
    cap = L.LZ4F_compressBound(0, ctypes.byref(p))
    dst = ctypes.create_string_buffer(cap)
    n = L.LZ4F_compressEnd(c, dst, cap, None)
    assert not L.LZ4F_isError(n), L.LZ4F_getErrorName(n)
    out += dst.raw[:n]
    # too small a destination is refused, a context without compressBegin too
    assert L.LZ4F_isError(L.LZ4F_compressUpdate(c, dst, cap, b"x" * 10, 10, None))
    L.LZ4F_freeCompressionContext(c)
    return bytes(out)


@pytest.mark.parametrize("kw", [
    dict(), dict(blockSizeID=5, contentChecksumFlag=1, blockChecksumFlag=1), dict(blockSizeID=4, blockMode=1, contentChecksumFlag=1),
    dict(blockSizeID=6, contentChecksumFlag=1, auto_flush=1), dict(blockSizeID=5, level=9, contentChecksumFlag=1),
    dict(blockSizeID=4, flush_every=3, blockChecksumFlag=1)])
def test_streaming_compression_context(L, oracle, datagen, kw):
    rng = random.Random(17)
    for n, pct in ((0, 50), (5, 50), (70000, 50), (1500000, 60)):
        data = datagen(n, pct, n % 5)
        frame = stream_compress(L, data, rng, **kw)
        assert frame[:4] == bytes.fromhex("04224d18")
        out = ctypes.create_string_buffer(n + 1)
        used = ctypes.c_size_t()
        r = oracle.lz4o_frame_decompress(out, n, frame, len(frame), ctypes.byref(used))
        assert r == n and used.value == len(frame) and out.raw[:n] == data, (kw, n)
        got, pos = decompress_frame(L, frame, n)
        assert got == data and pos == len(frame)
    # a declared content size must be honoured (lz4frame.c:1242-1245)
    data = datagen(100000, 50, 1)
    assert stream_compress(L, data, rng, contentSize=100000)[4] & 8
    with pytest.raises(AssertionError):
        stream_compress(L, data, rng, contentSize=99999)
    # linked streaming frames use the history: smaller than the same content in independent blocks
    data = datagen(1 << 20, 60, 3)
    assert len(stream_compress(L, data, rng, blockSizeID=4)) < len(stream_compress(L, data, rng, blockSizeID=4, blockMode=1))


def test_reference_cli_accepts_our_frames(L, datagen, tmp_path):
    """ours LZ4F_compressFrame / streaming context -> the reference CLI (`lz4 -t`, `lz4 -dc | cmp`).
    Needs the reference binary built into oracle/_ref (it travels with the snapshot)."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "lz4")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/lz4 not shipped")
    rng = random.Random(5)
    data = datagen(2500000, 60, 8)
    frames = {
        "one_shot_linked_cs": compress_frame(L, data, blockSizeID=5, contentChecksumFlag=1),
        "one_shot_indep_bx": compress_frame(L, data, blockSizeID=4, blockMode=1, blockChecksumFlag=1, contentSize=1),
        "one_shot_hc9": compress_frame(L, data, level=9, blockSizeID=6, contentChecksumFlag=1),
        "stream_linked": stream_compress(L, data, rng, blockSizeID=4, contentChecksumFlag=1),
        "stream_hc_flush": stream_compress(L, data, rng, level=9, blockSizeID=5, flush_every=2, blockChecksumFlag=1),
    }
    for name, frame in frames.items():
        f = tmp_path / (name + ".lz4")
        f.write_bytes(frame)
        t = subprocess.run([exe, "-t", str(f)], capture_output=True, text=True)
        assert t.returncode == 0, (name, t.stderr)
        d = subprocess.run([exe, "-dc", str(f)], capture_output=True)
        assert d.returncode == 0 and d.stdout == data, name


def _dict_frame(L, dictionary, body, **kw):
    st, vp = ctypes.c_size_t, ctypes.c_void_p
    L.LZ4F_createCDict.restype = vp
    L.LZ4F_createCDict.argtypes = [ctypes.c_char_p, st]
    L.LZ4F_freeCDict.argtypes = [vp]
    L.LZ4F_createCompressionContext.restype = st
    L.LZ4F_createCompressionContext.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
    L.LZ4F_freeCompressionContext.argtypes = [vp]
    L.LZ4F_compressFrame_usingCDict.restype = st
    L.LZ4F_compressFrame_usingCDict.argtypes = [vp, ctypes.c_char_p, st, ctypes.c_char_p, st, vp, ctypes.POINTER(Prefs)]
    p = Prefs()
    for k, v in kw.items():
        setattr(p.frameInfo, k, v)
    cd = L.LZ4F_createCDict(dictionary, len(dictionary))
    c = vp()
    assert cd and L.LZ4F_createCompressionContext(ctypes.byref(c), 100) == 0
    cap = L.LZ4F_compressFrameBound(len(body), ctypes.byref(p))
    dst = ctypes.create_string_buffer(cap)
    n = L.LZ4F_compressFrame_usingCDict(c, dst, cap, body, len(body), cd, ctypes.byref(p))
    assert not L.LZ4F_isError(n), L.LZ4F_getErrorName(n)
    L.LZ4F_freeCompressionContext(c); L.LZ4F_freeCDict(cd)
    return dst.raw[:n]


@pytest.mark.parametrize("mode", [0, 1])          # linked, independent blocks
def test_a_dictionary_counts_for_one_frame_and_after_get_frame_info(L, datagen, mode):
    """lz4frame.c:1327-1335, 2070-2083: LZ4F_decompress_usingDict's dictionary is taken while the frame has not started - also
    after LZ4F_getFrameInfo consumed the header (the way to read the dictID first) - and is forgotten when the frame ends: a
    later frame decoded with plain LZ4F_decompress on the same context must not see it."""
    st, vp = ctypes.c_size_t, ctypes.c_void_p
    L.LZ4F_decompress_usingDict.restype = st
    L.LZ4F_decompress_usingDict.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(st), ctypes.c_char_p, ctypes.POINTER(st), ctypes.c_char_p, st, vp]
    dictionary = datagen(40000, 60, 21)
    body = dictionary[5000:30000] + datagen(30000, 60, 22) + dictionary[100:9000]
    frame = _dict_frame(L, dictionary, body, blockSizeID=4, blockMode=mode)
    plain = compress_frame(L, body, blockSizeID=4, blockMode=mode)
    assert len(frame) < len(plain) - 5000                      # the dictionary was used

    def run(d, fr, use_dict, skip=0):
        out, pos = bytearray(), skip
        for _ in range(1000):
            dst = ctypes.create_string_buffer(200000)
            dsz, ssz = st(len(dst)), st(len(fr) - pos)
            if use_dict:
                r = L.LZ4F_decompress_usingDict(d, dst, ctypes.byref(dsz), fr[pos:], ctypes.byref(ssz), dictionary, len(dictionary), None)
            else:
                r = L.LZ4F_decompress(d, dst, ctypes.byref(dsz), fr[pos:], ctypes.byref(ssz), None)
            if L.LZ4F_isError(r):
                return None
            out += dst.raw[:dsz.value]; pos += ssz.value
            if r == 0:
                return bytes(out)
        return None

    d = vp()
    assert L.LZ4F_createDecompressionContext(ctypes.byref(d), 100) == 0
    # the header first (LZ4F_getFrameInfo consumes it), then the dictionary
    info, used = FrameInfo(), st(len(frame))
    assert not L.LZ4F_isError(L.LZ4F_getFrameInfo(d, ctypes.byref(info), frame, ctypes.byref(used)))
    assert run(d, frame, True, skip=used.value) == body
    # the same frame again on the same context, without the dictionary: its matches reach before the output
    assert run(d, frame, False) is None
    # ... and the context is usable afterwards
    assert run(d, frame, True) == body
    assert run(d, plain, False) == body
    L.LZ4F_freeDecompressionContext(d)


def test_frames_from_many_threads_at_once(L, oracle, datagen):
    """The frame entry points keep their device staging and stream per calling thread (csrc/lz4frame_api.c): eight threads compress
    and decode different frames at the same time - linked and independent blocks, with and without checksums - and every
    frame is what it is when made alone."""
    import threading
    kws = [dict(blockSizeID=4), dict(blockSizeID=5, blockMode=1, contentChecksumFlag=1), dict(blockSizeID=4, blockChecksumFlag=1), dict(blockSizeID=6, contentChecksumFlag=1)]
    jobs = [(datagen(700000 + 100000 * k, 60, 30 + k), kws[k % 4]) for k in range(8)]
    alone = [compress_frame(L, d, **kw) for d, kw in jobs]
    errs, frames = [], [None] * len(jobs)

    def work(k):
        try:
            d, kw = jobs[k]
            for _ in range(3):
                f = compress_frame(L, d, **kw)
                out, used = decompress_frame(L, f, len(d))
                assert out == d and used == len(f)
            frames[k] = f
        except Exception as e:                                        # noqa: BLE001
            errs.append((k, repr(e)))
    ts = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert frames == alone


@pytest.mark.parametrize("kw", [dict(blockSizeID=7, blockMode=0, contentChecksumFlag=1), dict(blockSizeID=4, blockMode=1),
                                dict(blockSizeID=6, blockMode=1, blockChecksumFlag=1, contentChecksumFlag=1)])
def test_large_frames_through_overlapping_batches(L, datagen, kw):
    """Frames of several batches (lz4frame_api.c: a batch that may decode to 8 MiB or more runs on a helper thread while the
    calling thread hands out the batch before and takes in the next): the content is the same whatever the caller's buffers
    are - the whole frame at once, pieces of up to 3 MB into destinations of up to 5 MB, and a destination so small that
    most calls return with a batch still in flight.  Compressible, incompressible (stored blocks, some of them handed on in
    pieces) and mixed content; other frames are compressed and decoded on the same thread in between (a batch on its helper
    thread has a staging area of its own)."""
    import random
    rng = random.Random(17)
    data = datagen(24 << 20, 60, 3) + os.urandom(9 << 20) + datagen(20 << 20, 90, 4) + os.urandom(70000) + datagen(3 << 20, 20, 5)
    frame = compress_frame(L, data, **kw)
    for take_max, dcap_max in ((1 << 30, 1 << 30), (3 << 20, 5 << 20), (9 << 20, 300000), (200000, 9 << 20)):
        d = ctypes.c_void_p()
        assert L.LZ4F_createDecompressionContext(ctypes.byref(d), 100) == 0
        out, pos = bytearray(), 0
        try:
            for _ in range(200000):
                take = min(len(frame) - pos, rng.randint(1, take_max))
                dst = ctypes.create_string_buffer(min(len(data) + 64, rng.randint(1, dcap_max)))
                dsz, ssz = ctypes.c_size_t(len(dst)), ctypes.c_size_t(take)
                r = L.LZ4F_decompress(d, dst, ctypes.byref(dsz), frame[pos:pos + take], ctypes.byref(ssz), None)
                assert not L.LZ4F_isError(r), L.LZ4F_getErrorName(r)
                out += dst.raw[:dsz.value]
                pos += ssz.value
                if r == 0:
                    break
                if dcap_max == 300000 and _ % 40 == 7:
                    # another frame function on the same thread while a batch of this decoder is on the device: it has its own staging area
                    other = data[(_ * 4096) % (1 << 20):][:3 << 20]
                    back = decompress_frame(L, compress_frame(L, other, blockSizeID=5), len(other))
                    assert back[0] == other
        finally:
            L.LZ4F_freeDecompressionContext(d)
        assert pos == len(frame) and bytes(out) == data, (kw, take_max, dcap_max, len(out))


@pytest.mark.parametrize("kw", [dict(blockSizeID=7, blockMode=0, contentChecksumFlag=1), dict(blockSizeID=6, blockMode=1, blockChecksumFlag=1)])
def test_large_batches_when_no_helper_thread_can_be_started(L, datagen, kw):
    """A large batch normally runs on a helper thread while the batch before it is handed out.  When that thread cannot be
    started (LZ4AMD_TEST_NO_BATCH_THREAD=1: the way pthread_create fails in a process that is out of threads) the batch is
    decoded on the calling thread - but only after every byte of the batch before has been handed out: decoding it at once
    would put its bytes where those still wait (they were silently lost: round 4's advisor finding).  Small destinations, so
    that a batch is pending whenever the next one is ready."""
    import random
    rng = random.Random(23)
    data = datagen(40 << 20, 60, 9) + os.urandom(5 << 20) + datagen(30 << 20, 90, 10)
    frame = compress_frame(L, data, **kw)
    for env in ("1", "0"):
        os.environ["LZ4AMD_TEST_NO_BATCH_THREAD"] = env
        try:
            for take_max, dcap_max in ((9 << 20, 300000), (1 << 30, 1 << 20), (40 << 20, 64 << 20)):
                d = ctypes.c_void_p()
                assert L.LZ4F_createDecompressionContext(ctypes.byref(d), 100) == 0
                out, pos = bytearray(), 0
                try:
                    for _ in range(200000):
                        take = min(len(frame) - pos, rng.randint(1, take_max))
                        dst = ctypes.create_string_buffer(min(len(data) + 64, rng.randint(1, dcap_max)))
                        dsz, ssz = ctypes.c_size_t(len(dst)), ctypes.c_size_t(take)
                        r = L.LZ4F_decompress(d, dst, ctypes.byref(dsz), frame[pos:pos + take], ctypes.byref(ssz), None)
                        assert not L.LZ4F_isError(r), L.LZ4F_getErrorName(r)
                        out += dst.raw[:dsz.value]
                        pos += ssz.value
                        if r == 0:
                            break
                finally:
                    L.LZ4F_freeDecompressionContext(d)
                assert pos == len(frame) and bytes(out) == data, (kw, env, take_max, dcap_max, len(out))
        finally:
            os.environ.pop("LZ4AMD_TEST_NO_BATCH_THREAD", None)


def test_empty_stored_blocks_with_checksums_between_large_batches(L, datagen):
    """tests/frametest.c splices empty stored blocks (header 0x80000000, then - with block checksums - the XXH32 of nothing)
    into frames.  One that arrives right behind a large batch, its checksum cut in two by the caller's chunking while the batch
    is still on the device, must not lose the checksum bytes already taken in (lz4frame_api.c: the piecewise path of stored blocks)."""
    import struct
    import random
    data = datagen(20 << 20, 60, 21)
    frame = compress_frame(L, data, blockSizeID=7, blockMode=1, blockChecksumFlag=1, contentChecksumFlag=1)
    # walk the blocks (header 7 bytes: no content size, no dictID), put an empty stored block behind every one of them
    pos, out, cuts = 7, bytearray(frame[:7]), []
    empty = struct.pack("<I", 0x80000000) + struct.pack("<I", 0x02CC5D05)
    while True:
        f = struct.unpack_from("<I", frame, pos)[0]
        if f == 0:
            break
        n = 4 + (f & 0x7FFFFFFF) + 4
        out += frame[pos:pos + n]
        cuts.append(len(out))
        out += empty
        pos += n
    out += frame[pos:]
    spliced = bytes(out)
    assert len(cuts) >= 5
    rng = random.Random(5)
    for trial in range(12):
        d = ctypes.c_void_p()
        assert L.LZ4F_createDecompressionContext(ctypes.byref(d), 100) == 0
        got, p = bytearray(), 0
        # input is handed over up to a point 5, 6 or 7 bytes into an empty block (its header and a piece of its checksum), then in small steps
        stops = sorted(c + rng.choice((4, 5, 6, 7)) for c in rng.sample(cuts, 3))
        try:
            for _ in range(400000):
                nxt = next((s for s in stops if s > p), len(spliced))
                take = min(nxt - p, rng.randint(1, 6 << 20)) if nxt > p else 0
                dst = ctypes.create_string_buffer(rng.choice((1 << 16, 300000, 5 << 20)))
                dsz, ssz = ctypes.c_size_t(len(dst)), ctypes.c_size_t(take)
                r = L.LZ4F_decompress(d, dst, ctypes.byref(dsz), spliced[p:p + take], ctypes.byref(ssz), None)
                assert not L.LZ4F_isError(r), (trial, L.LZ4F_getErrorName(r))
                got += dst.raw[:dsz.value]
                p += ssz.value
                if r == 0:
                    break
                if p in stops and ssz.value == 0 and dsz.value == 0:
                    stops.remove(p)                               # nothing moves any more at the stop: go on
        finally:
            L.LZ4F_freeDecompressionContext(d)
        assert p == len(spliced) and bytes(got) == data, (trial, len(got))
