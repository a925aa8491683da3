"""LZ4F_decompress as a streaming state machine (lz4frame.c:1613-2060), exercised WITHOUT a GPU: frames whose blocks
are all stored (high bit of the block size set, lz4frame.c:1758-1830) never visit the device, so the container logic
- item-wise input, bounded buffering, progressive output, size hints, getFrameInfo consuming the header, skippable
frames, checksums - runs in the CPU suite.  Where oracle/_ref/liblz4_ref.so exists the very same call sequence is
replayed on the real reference and consumed / produced counts are compared call by call."""
import ctypes
import os
import random
import struct

import pytest
import xxhash

from test_gpu_frame import FrameInfo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BSIZE = {4: 65536, 5: 262144, 6: 1 << 20, 7: 4 << 20}


def stored_frame(data, bsid=4, linked=False, content_checksum=True, content_size=False, block_size=None, block_checksum=False):
    flg = 0x40 | (0 if linked else 0x20) | (4 if content_checksum else 0) | (8 if content_size else 0) | (0x10 if block_checksum else 0)
    desc = bytes([flg, bsid << 4]) + (struct.pack("<Q", len(data)) if content_size else b"")
    out = bytearray(struct.pack("<I", 0x184D2204) + desc + bytes([(xxhash.xxh32(desc).intdigest() >> 8) & 0xFF]))
    bs = block_size or BSIZE[bsid]
    for o in range(0, len(data), bs):
        blk = data[o:o + bs]
        out += struct.pack("<I", len(blk) | 0x80000000) + blk + (struct.pack("<I", xxhash.xxh32(blk).intdigest()) if block_checksum else b"")
    out += struct.pack("<I", 0)
    if content_checksum:
        out += struct.pack("<I", xxhash.xxh32(data).intdigest())
    return bytes(out)


def bind(lib):
    st, vp = ctypes.c_size_t, ctypes.c_void_p
    lib.LZ4F_isError.argtypes = [st]
    lib.LZ4F_getErrorName.restype = ctypes.c_char_p
    lib.LZ4F_getErrorName.argtypes = [st]
    lib.LZ4F_createDecompressionContext.restype = st
    lib.LZ4F_createDecompressionContext.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
    lib.LZ4F_freeDecompressionContext.argtypes = [vp]
    lib.LZ4F_resetDecompressionContext.argtypes = [vp]
    lib.LZ4F_decompress.restype = st
    lib.LZ4F_decompress.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(st), ctypes.c_char_p, ctypes.POINTER(st), vp]
    lib.LZ4F_getFrameInfo.restype = st
    lib.LZ4F_getFrameInfo.argtypes = [vp, ctypes.POINTER(FrameInfo), ctypes.c_char_p, ctypes.POINTER(st)]
    return lib


@pytest.fixture(scope="module")
def L():
    import lz4_amd
    return bind(lz4_amd.lib())


@pytest.fixture(scope="module")
def R():
    p = os.path.join(ROOT, "oracle", "_ref", "liblz4_ref.so")
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/liblz4_ref.so not built")
    return bind(ctypes.CDLL(p))


class Dctx:
    def __init__(self, lib):
        self.lib, self.d = lib, ctypes.c_void_p()
        assert lib.LZ4F_createDecompressionContext(ctypes.byref(self.d), 100) == 0

    def call(self, src, dcap):
        dst = ctypes.create_string_buffer(max(dcap, 1))
        dsz, ssz = ctypes.c_size_t(dcap), ctypes.c_size_t(len(src))
        r = self.lib.LZ4F_decompress(self.d, dst, ctypes.byref(dsz), src, ctypes.byref(ssz), None)
        return r, ssz.value, dst.raw[:dsz.value]

    def info(self, src):
        fi, ssz = FrameInfo(), ctypes.c_size_t(len(src))
        r = self.lib.LZ4F_getFrameInfo(self.d, ctypes.byref(fi), src, ctypes.byref(ssz))
        return r, ssz.value, fi

    def close(self):
        self.lib.LZ4F_freeDecompressionContext(self.d)


def drive(lib, frame, schedule):
    """Feed `frame` in the pieces / destination sizes of `schedule` ((take, dcap) pairs, cycled); returns the output and
    the per-call trace (consumed, produced, finished)."""
    d, out, pos, trace = Dctx(lib), bytearray(), 0, []
    try:
        for k in range(200000):
            take, dcap = schedule[k % len(schedule)]
            r, used, got = d.call(frame[pos:pos + take], dcap)
            assert not lib.LZ4F_isError(r), lib.LZ4F_getErrorName(r)
            assert used <= min(take, len(frame) - pos)
            out += got
            pos += used
            trace.append((used, len(got), r == 0))
            if r == 0:
                return bytes(out), pos, trace
        raise AssertionError("no end of frame")
    finally:
        d.close()


def test_output_is_progressive_and_input_is_taken_item_by_item(L):
    rng = random.Random(1)
    data = bytes(rng.getrandbits(8) for _ in range(200000))
    frame = stored_frame(data, bsid=4)
    d, pos, out = Dctx(L), 0, bytearray()
    # header alone: consumed, nothing produced, the hint asks for the next block header
    r, used, got = d.call(frame[:7], 1000)
    assert (r, used, got) == (4, 7, b"")
    pos = 7
    # first block header + half of the block: a stored block needs no buffer - its bytes come out as they arrive (lz4frame.c:1790-1830),
    # hint = what the block misses + next header
    r, used, got = d.call(frame[pos:pos + 4 + 30000], 1 << 20)
    assert used == 30004 and got == data[:30000] and r == 65536 - 30000 + 4
    pos += used
    # the rest of block 0 and a bit of block 1: the rest of block 0 comes out, and the 100 bytes of block 1 that are there
    r, used, got = d.call(frame[pos:pos + 35536 + 4 + 100], 1 << 20)
    assert used == 35536 + 4 + 100 and got == data[30000:65636] and r == 65536 - 100 + 4
    pos += used
    # a small destination: the decoder holds the bytes and takes no input until they are delivered
    r, used, got = d.call(frame[pos:pos + 65436], 1 << 20)
    assert used == 65436 and got == data[65636:131072]
    pos += used
    r, used, got = d.call(frame[pos:], 1000)               # block 2, the short block 3, end mark, checksum are all there: whole blocks go as a batch
    assert got == data[131072:132072] and r != 0
    pos += used
    out = bytearray(data[:132072])
    while r != 0:
        before = pos
        r, used, got = d.call(frame[pos:], 50000)
        assert not L.LZ4F_isError(r)
        pos += used
        out += got
        assert used or got
    assert bytes(out) == data and pos == len(frame)
    d.close()


@pytest.mark.parametrize("kw", [dict(bsid=4), dict(bsid=5, linked=True), dict(bsid=4, content_checksum=False, content_size=True),
                                dict(bsid=7, block_size=70001), dict(bsid=4, block_size=1), dict(bsid=4, block_checksum=True), dict(bsid=5, linked=True, block_checksum=True)])
def test_any_chunking_gives_the_content(L, kw):
    rng = random.Random(7)
    small = kw.get("block_size") == 1
    for n, schedules in ((3000, ([(1, 1)], [(1, 1 << 20)], [(1 << 24, 1)], [(1 << 24, 1 << 24)])),
                         (700001, ([(1 << 24, 4097)], [(rng.randint(1, 100000), rng.randint(1, 100000)) for _ in range(64)],
                                   [(1 << 24, 1 << 24)]))):
        if small and n > 3000:
            continue
        data = bytes(rng.getrandbits(8) for _ in range(1000)) * (n // 1000) + b"tail" * (n % 1000 > 0)
        frame = stored_frame(data, **kw)
        for schedule in schedules:
            out, pos, _ = drive(L, frame + b"following bytes", schedule)
            assert out == data and pos == len(frame)


def test_hint_driven_reading_reaches_the_end(L):
    """A caller that reads exactly what the previous call asked for (the lz4io.c pattern) finishes the frame and
    never reads past it."""
    data = os.urandom(300000)
    frame = stored_frame(data, bsid=4, content_size=True) + b"XXXX"
    d, pos, out, want = Dctx(L), 0, bytearray(), 7
    for _ in range(100):
        r, used, got = d.call(frame[pos:pos + want], 1 << 20)
        assert not L.LZ4F_isError(r) and used == min(want, len(frame) - pos)
        pos += used
        out += got
        if r == 0:
            break
        want = r
    assert r == 0 and bytes(out) == data and pos == len(frame) - 4
    d.close()


def test_get_frame_info_consumes_the_header(L):
    data = os.urandom(1000)
    frame = stored_frame(data, bsid=6, linked=True, content_size=True)
    d = Dctx(L)
    r, used, fi = d.info(frame[:6])                                    # not whole: an error, nothing consumed
    assert L.LZ4F_isError(r) and used == 0 and b"frameHeader_incomplete" in L.LZ4F_getErrorName(r)
    r, used, fi = d.info(frame)
    assert r == 4 and used == 15
    assert (fi.blockSizeID, fi.blockMode, fi.contentChecksumFlag, fi.contentSize) == (6, 0, 1, 1000)
    r, used2, fi2 = d.info(frame[used:])                               # again: same answer, nothing consumed
    assert not L.LZ4F_isError(r) and used2 == 0 and fi2.contentSize == 1000
    r, used3, got = d.call(frame[used:], 5000)                          # decoding goes on after the header
    assert r == 0 and used + used3 == len(frame) and got == data
    # half a header through LZ4F_decompress, then getFrameInfo: refused like lz4frame.c:1478
    r, used, got = d.call(frame[:5], 10)
    assert used == 5 and r == 6                                         # 2 to the minimal header + a block header
    r, used, fi = d.info(frame[5:])
    assert L.LZ4F_isError(r) and used == 0 and b"alreadyStarted" in L.LZ4F_getErrorName(r)
    d.close()


def test_skippable_frames_errors_and_context_reuse(L):
    data = os.urandom(70000)
    frame = stored_frame(data, bsid=4, content_size=True)
    skip = struct.pack("<II", 0x184D2A53, 11) + b"hello world"
    out, pos, trace = drive(L, skip + frame, [(5, 100)])
    assert out == b"" and pos == len(skip)                              # a skippable frame ends like a frame
    d = Dctx(L)
    r, used, got = d.call(skip + frame, 1 << 20)
    assert (r, used, got) == (0, len(skip), b"")
    r, used, got = d.call(frame, 1 << 20)                               # same context, next frame
    assert (r, used, got) == (0, len(frame), data)
    bad = bytearray(frame); bad[-1] ^= 1
    r, used, got = d.call(bytes(bad), 1 << 20)
    assert L.LZ4F_isError(r) and b"contentChecksum" in L.LZ4F_getErrorName(r)
    r, used, got = d.call(frame, 1 << 20)                               # an error leaves the context reusable
    assert (r, used, got) == (0, len(frame), data)
    short = stored_frame(data[:-1], bsid=4)
    wrong = frame[:15] + short[7:]                                      # header promises 70000 bytes, 69999 follow
    r, used, got = d.call(wrong, 1 << 20)
    assert L.LZ4F_isError(r) and b"frameSize_wrong" in L.LZ4F_getErrorName(r)
    big = bytearray(frame); big[15:19] = struct.pack("<I", 0x80000000 | 65537)
    r, used, got = d.call(bytes(big), 1 << 20)
    assert L.LZ4F_isError(r) and b"maxBlockSize_invalid" in L.LZ4F_getErrorName(r)
    d.close()


def test_a_stored_block_with_a_wrong_checksum_is_refused_whole_or_in_pieces(L):
    """Block checksums of stored blocks (lz4frame.c:1878) are verified on the host: over the batch when the blocks arrive whole,
    over the pieces when a block is handed on as it arrives."""
    rng = random.Random(3)
    data = bytes(rng.getrandbits(8) for _ in range(150000))
    frame = bytearray(stored_frame(data, bsid=4, block_checksum=True))
    frame[7 + 4 + 70000] ^= 1                                        # a byte of block 1
    for take in (1 << 24, 5000):
        d, pos, err = Dctx(L), 0, None
        for _ in range(1000):
            r, used, got = d.call(bytes(frame[pos:pos + take]), 1 << 20)
            pos += used
            if L.LZ4F_isError(r):
                err = L.LZ4F_getErrorName(r)
                break
            if r == 0:
                break
        assert err == b"ERROR_blockChecksum_invalid", (take, err)
        d.close()


def test_memory_is_bounded_by_a_batch_not_by_the_frame(L):
    """512 MiB of stored 4 MiB blocks streamed through a 1 MiB window: the resident set must not grow with the frame."""
    import resource
    blk = os.urandom(4 << 20)
    head = stored_frame(b"", bsid=7, content_checksum=False)[:7]
    piece = struct.pack("<I", len(blk) | 0x80000000) + blk
    d, total = Dctx(L), 0
    r, used, got = d.call(head, 0)
    assert used == 7
    base = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    for k in range(128):
        pos = 0
        while pos < len(piece):
            r, used, got = d.call(piece[pos:pos + (1 << 20)], 1 << 20)
            assert not L.LZ4F_isError(r)
            pos += used
            total += len(got)
    while total < 128 * len(blk):
        r, used, got = d.call(b"", 1 << 20)
        total += len(got)
    r, used, got = d.call(struct.pack("<I", 0), 10)
    assert r == 0 and total == 128 * len(blk)
    assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - base < 64 << 10        # KiB: < 64 MiB for a 512 MiB frame
    d.close()


def test_same_call_sequence_on_the_reference(L, R):
    """Replay identical call sequences on the real lz4frame.c: the content, the bytes consumed in total and the call
    that reports the end of the frame agree; so do the per-call consumed counts when every call offers exactly the
    bytes the previous one asked for, and the whole per-call trace when the input comes in pieces smaller than a block."""
    rng = random.Random(11)
    data = bytes(rng.getrandbits(8) for _ in range(5000)) * 60
    for kw in (dict(bsid=4), dict(bsid=5, linked=True, content_size=True), dict(bsid=4, content_checksum=False)):
        frame = stored_frame(data, **kw)
        for schedule in ([(rng.randint(1, 90000), rng.randint(1, 90000)) for _ in range(50)], [(1 << 24, 1 << 24)], [(97, 70000)], [(3000, 1000)], [(20000, 30000)]):
            a, pa, ta = drive(L, frame, schedule)
            b, pb, tb = drive(R, frame, schedule)
            assert a == b == data and pa == pb == len(frame)
            # input offered in pieces smaller than a block: a stored block is handed on as it arrives, call by call like the reference
            # (lz4frame.c:1790-1830) - the two traces of (consumed, produced, finished) are the same
            if max(t for t, _ in schedule) < 60000:
                assert ta == tb, schedule
        # hint-driven on both
        seqs = []
        for lib in (L, R):
            d, pos, want, seq, out = Dctx(lib), 0, 7, [], bytearray()
            for _ in range(1000):
                r, used, got = d.call(frame[pos:pos + want], 1 << 20)
                assert not lib.LZ4F_isError(r)
                pos += used; out += got; seq.append(used)
                if r == 0:
                    break
                want = r
            assert bytes(out) == data and pos == len(frame)
            d.close()
            seqs.append(seq)
        assert seqs[0] == seqs[1]
