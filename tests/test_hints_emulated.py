"""Entry-point tables ("hints", include/lz4amd.h) on the CPU SIMT interpreter: the tables lz4amd_k_compress writes are
checked row by row against the block's real token chain, the decoder's PARSER path is driven with them and with tables
made here for blocks of ANY origin (reference-compressed, HC, hand-built sequence lists), and with tables that lie.
Test infrastructure: the -m gpu twin of this file is tests/test_gpu_hints.py."""
import ctypes
import os
import random
import struct

import pytest

import test_kernels_emulated as tk

from hint_format import MAGIC, hint_bytes, pack_table, unpack_row, set_row, header, is_valid, row_offset, table_bytes
CANARY = 0xEE


EVERY = 16                     # lz4amd_k_compress's rows are at most 16 sequences apart (LZ4AMD_HINT_EVERY_MAX; 8 and fewer on data of 32 bytes per sequence and more)


def token_chain(comp):
    """[(token position, output position of the sequence's first literal)] of a legal block, and its decoded size."""
    p = o = 0
    out = []
    while True:
        out.append((p, o))
        b = comp[p]; ll = b >> 4; q = p + 1
        if ll == 15:
            while True:
                x = comp[q]; q += 1; ll += x
                if x != 255:
                    break
        if q + ll >= len(comp):
            assert q + ll == len(comp)
            return out, o + ll
        m = q + ll; ml = b & 15; nx = m + 2
        if ml == 15:
            while True:
                x = comp[nx]; nx += 1; ml += x
                if x != 255:
                    break
        o += ll + ml + 4; p = nx


def make_table(comp, every=EVERY, by_bytes=0):
    """A valid table for any legal block: a row per `every` sequences, or (by_bytes) a row per `by_bytes` bytes of output - row r
    then names the first sequence that starts at or behind byte by_bytes * r (the last sequence when there is none)."""
    ch, n = token_chain(comp)
    rows = []
    if by_bytes:
        j = 0
        for r in range((n + by_bytes - 1) // by_bytes):
            want = r * by_bytes
            while j + 1 < len(ch) and ch[j][1] < want:
                j += 1
            k = j if ch[j][1] >= want else len(ch) - 1
            rows.append((ch[k][0], ch[k][1], k))
        if rows:
            rows[0] = (0, 0, 0)
            for r in range(1, len(rows)):       # never decreasing
                if rows[r] < rows[r - 1]:
                    rows[r] = rows[r - 1]
    else:
        rows = [(ch[k][0], ch[k][1], k) for k in range(0, len(ch), every)]
    if not rows:
        rows = [(0, 0, 0)]
    return pack_table(n, len(comp), len(ch), rows)


def check_table(comp, table, n):
    """a table lz4amd_k_compress wrote: every row is a sequence of the block's real token chain, rows are at most 8 sequences
    apart, the block's last sequence has a row, the first row carries the number of rows"""
    magic, osz, csz, nseq, nrows = header(table)
    assert (magic, osz, csz) == (MAGIC, n, len(comp))
    ch, total = token_chain(comp)
    assert total == n and nseq == len(ch)
    assert struct.unpack_from("<3I", table, 20) == (0, 0, 0) and 1 <= nrows <= nseq and table_bytes(nrows) <= len(table)
    assert unpack_row(table, 0) == (0, 0, 0)
    prev = 0
    for r in range(1, nrows):
        tok, out, olo = unpack_row(table, r)
        ordn = prev + ((olo - prev) & 0xFF)                          # (the rows carry the count's low 8 bits: neighbours are at most 255 apart)
        assert prev < ordn <= prev + EVERY and ordn < nseq and (tok, out) == ch[ordn], (r, tok, out, ordn)
        prev = ordn
    assert prev == nseq - 1 or nrows == 1 and nseq <= EVERY + 1, (prev, nseq)
    assert unpack_row(table, nrows) == (len(comp), n, nseq & 0xFF)
    return nrows


def emu_compress_tables(emu, datas, accel=1):
    n = len(datas)
    caps = [len(d) + len(d) // 255 + 16 for d in datas]
    srcs = [ctypes.create_string_buffer(d, len(d)) if d else ctypes.create_string_buffer(1) for d in datas]
    dsts = [ctypes.create_string_buffer(c + 64) for c in caps]
    stride = max(hint_bytes(len(d)) for d in datas)
    hraw = ctypes.create_string_buffer(stride * n + 16)
    hbase = (ctypes.addressof(hraw) + 15) & ~15
    sp = (ctypes.c_void_p * n)(*[ctypes.addressof(s) for s in srcs])
    dp = (ctypes.c_void_p * n)(*[ctypes.addressof(d) for d in dsts])
    ss = (ctypes.c_int32 * n)(*[len(d) for d in datas]); dc = (ctypes.c_int32 * n)(*caps); res = (ctypes.c_int32 * n)()
    emu.emu_compress_batch_hints(sp, ss, dp, dc, res, n, 0, None, ctypes.c_void_p(hbase), ctypes.c_uint64(stride), accel)
    off = hbase - ctypes.addressof(hraw)
    return [dsts[i].raw[:res[i]] for i in range(n)], [hraw.raw[off + i * stride: off + (i + 1) * stride] for i in range(n)]


def emu_decompress_tables(emu, blocks, caps, tables, salign=0, align=0, prefixes=None, make=None):
    """-> ([(result, bytes)], blocks decoded from their table, tables rejected); canaries around every output.
    make (a list): the decoder writes the tables of blocks that have none; the list receives [tables afterwards, tables made]."""
    n = len(blocks)
    srcs = [ctypes.create_string_buffer(len(b) + 64 + salign) for b in blocks]
    sptr = lambda buf: ((ctypes.addressof(buf) + 15) & ~15) + salign
    for s, b in zip(srcs, blocks):
        ctypes.memset(s, 0xA5, len(s)); ctypes.memmove(sptr(s), b, len(b))
    pres = prefixes or [b""] * n
    dsts = [ctypes.create_string_buffer(len(p) + max(c, 0) + 96) for p, c in zip(pres, caps)]
    ptr = lambda i: ((ctypes.addressof(dsts[i]) + 15) & ~15) + align + len(pres[i])
    for i, d in enumerate(dsts):
        ctypes.memset(d, CANARY, len(d))
        ctypes.memmove(ptr(i) - len(pres[i]), pres[i], len(pres[i]))
    stride = max(16 * ((len(t) + 15) // 16) for t in tables)
    hraw = ctypes.create_string_buffer(stride * n + 16)
    hbase = (ctypes.addressof(hraw) + 15) & ~15
    for i, t in enumerate(tables):
        ctypes.memmove(hbase + i * stride, t, len(t))
    sp = (ctypes.c_void_p * n)(*[sptr(s) for s in srcs]); dp = (ctypes.c_void_p * n)(*[ptr(i) for i in range(n)])
    ss = (ctypes.c_int32 * n)(*[len(b) for b in blocks]); dc = (ctypes.c_int32 * n)(*caps); res = (ctypes.c_int32 * n)()
    pre = (ctypes.c_int32 * n)(*[len(p) for p in pres]) if prefixes else None
    stats = (ctypes.c_uint32 * 3)()
    emu.emu_decompress_batch_hints_make(sp, ss, dp, dc, res, n, 0, pre, ctypes.c_void_p(hbase), ctypes.c_uint64(stride), stats, 1 if make is not None else 0)
    if make is not None:
        off = hbase - ctypes.addressof(hraw)
        make[:] = [[hraw.raw[off + i * stride: off + (i + 1) * stride] for i in range(n)], stats[2]]
    outs = []
    for i in range(n):
        off = ptr(i) - ctypes.addressof(dsts[i])
        raw = dsts[i].raw
        cap = max(caps[i], 0)
        assert raw[off + cap:off + cap + 32] == bytes([CANARY]) * 32, f"block {i}: wrote past dst[cap]"
        assert raw[off - len(pres[i]):off] == pres[i] and raw[:off - len(pres[i])] == bytes([CANARY]) * (off - len(pres[i]))
        outs.append((res[i], raw[off:off + max(res[i], 0)]))
    return outs, stats[0], stats[1]


def foreign_cases(ocodec, reflib, datagen):
    """(decoded bytes, block) of every decoder corpus of test_kernels_emulated: reference-fast, reference-HC and hand-built blocks"""
    cases = []
    specs = [(65536, 50, 0), (100, 50, 1), (13, 50, 0), (12, 50, 0), (200000, 60, 2), (1 << 20, 60, 3), (300000, 90, 4), (50000, 0, 5),
             (1, 50, 0), (65547, 50, 1), (131073, 60, 1), (300000, 20, 6)]
    datas = [datagen(*s) for s in specs]
    datas += tk._region_index_corpus() + tk._periodic_corpus()[::2] + tk._field_length_corpus()[:2]
    datas += [b"\x00" * 300000, b"abcd" * 70000, os.urandom(70000), b"x" * 64, b"x" * 65, bytes(2 << 20)]
    for d in datas:
        cases.append((d, ocodec.compress(d)[1]))
    for d in [random.Random(11).randbytes(1 << 20), b"0123456789abcdef" * (1 << 16), datas[4], tk._field_length_corpus()[0]]:
        cap = len(d) + len(d) // 255 + 16
        cb = ctypes.create_string_buffer(cap)
        n = reflib.LZ4_compress_HC(d, cb, len(d), cap, 9)
        cases.append((d, cb.raw[:n]))
    rnd = random.Random(2024)
    for target in [1, 30, 300, 1000, 1023, 1024, 1025, 5000, 20000, 70000, 150000] + [rnd.randrange(10, 60000) for _ in range(8)]:
        c, d = tk._random_legal_block(rnd, target)
        cases.append((d, c))
    return cases


@pytest.fixture(scope="module")
def foreign(ocodec, reflib, datagen):
    return foreign_cases(ocodec, reflib, datagen)


def test_tables_made_for_foreign_blocks_decode_every_corpus(emu, foreign):
    blocks = [c for _, c in foreign]
    wants = [d for d, _ in foreign]
    for every, by_bytes, sal in ((8, 0, 0), (8, 0, 5), (1, 0, 0), (50, 0, 0), (0, 512, 0), (0, 3000, 3), (2000, 0, 0)):
        tables = [make_table(c, every, by_bytes) for c in blocks]
        outs, used, rejected = emu_decompress_tables(emu, blocks, [len(d) for d in wants], tables, salign=sal)
        for d, (r, o) in zip(wants, outs):
            assert r == len(d) and o == d, (every, by_bytes, sal, len(d))
        # (rows more than 255 sequences apart - every 2000 sequences; every 3000 bytes of a block of 6-byte sequences - cannot be said in the rows' 8 bits:
        #  the count the parser keeps then does not match what the lanes walk, and such a table is rejected, not wrong)
        assert used + rejected == len(blocks) and (rejected == 0 or every == 2000 or by_bytes == 3000) and used >= 40, (every, by_bytes, used, rejected)


def test_compressor_tables_name_real_sequences_and_are_used(emu, ocodec, datagen):
    specs = [(200000, 60, 2), (65536, 50, 0), (1 << 20, 60, 3), (300000, 90, 4), (50000, 0, 5), (100, 50, 1), (13, 50, 0), (5000, 20, 1),
             (131073, 60, 1), (4 << 20, 60, 0), (700000, 20, 3), (1024, 60, 1), (1025, 60, 1), (2048, 90, 2), (65535, 60, 1)]
    datas = [datagen(*s) for s in specs] + [b"\x00" * 300000, b"abcd" * 70000, os.urandom(70000), b"a" * 40000 + os.urandom(3000) + b"a" * 40000,
                                            os.urandom(1 << 20) + b"q" * 100000 + os.urandom(50000)] + tk._region_index_corpus()[2:5]
    comps, tables = emu_compress_tables(emu, datas)
    plain = tk.emu_compress(emu, datas, align=0)
    for d, c, t, (pr, pc) in zip(datas, comps, tables, plain):
        assert c == pc                                            # the block is byte for byte what it is without the table
        ro, o = ocodec.decompress(c, len(d))
        assert ro == len(d) and o == d
        check_table(c, t, len(d))
    for sal in (0, 9):
        outs, used, rejected = emu_decompress_tables(emu, comps, [len(d) for d in datas], tables, salign=sal)
        for d, (r, o) in zip(datas, outs):
            assert r == len(d) and o == d
        assert used == len(datas) and rejected == 0
    # more room than the block needs is fine; less is the same failure as without a table
    outs, used, _ = emu_decompress_tables(emu, comps[:3], [len(d) + 100 for d in datas[:3]], tables[:3])
    assert [r for r, _ in outs] == [len(d) for d in datas[:3]] and used == 3
    outs, used, _ = emu_decompress_tables(emu, comps[:3], [len(d) - 1 for d in datas[:3]], tables[:3])
    assert all(r < 0 for r, _ in outs) and used == 0
    # a failed or empty compression leaves no valid table
    comps2, tables2 = emu_compress_tables(emu, [b"", datas[0]])
    assert not is_valid(tables2[0]) and is_valid(tables2[1])


def test_tables_that_lie_only_cost_time(emu, ocodec, datagen):
    rnd = random.Random(77)
    d = datagen(300000, 60, 5)
    c = ocodec.compress(d)[1]
    good = make_table(c)
    other = make_table(ocodec.compress(datagen(300000, 60, 6))[1])
    nrows = header(good)[4]
    blocks, tables = [], []
    for t in range(60):
        bad = bytearray(good)
        kind = t % 6
        if kind == 0:                                              # random words anywhere behind the header
            for _ in range(rnd.randint(1, 4)):
                struct.pack_into("<I", bad, 32 + 4 * rnd.randrange(2 * (nrows - 1)), rnd.randrange(1 << 22))
        elif kind == 1:                                            # a row moved a little: still inside the block, not on the chain
            r = rnd.randrange(1, nrows - 1)
            tok, out, olo = unpack_row(bad, r)
            set_row(bad, r, tok + rnd.choice((1, 2, 3)), out + rnd.choice((0, 1)), olo)
        elif kind == 2:                                            # a true token with the wrong output position / count
            r = rnd.randrange(1, nrows - 1)
            tok, out, olo = unpack_row(bad, r)
            set_row(bad, r, tok, out + (1 if t % 2 else 0), olo + (0 if t % 2 else 1))
        elif kind == 3:                                            # header says something else
            struct.pack_into("<I", bad, 4 * rnd.randrange(1, 4), rnd.randrange(1 << 20))
        elif kind == 4:                                            # another block's table
            bad = bytearray(other[:len(good)].ljust(len(good), b"\0"))
        else:                                                      # noise
            bad = bytearray(rnd.randbytes(len(good)))
            if t % 2:
                struct.pack_into("<8I", bad, 0, *struct.unpack_from("<8I", good, 0))
        blocks.append(c); tables.append(bytes(bad))
    outs, used, rejected = emu_decompress_tables(emu, blocks, [len(d)] * len(blocks), tables)
    for r, o in outs:
        assert r == len(d) and o == d
    assert used + rejected >= 30 and rejected >= 25               # (tables with a broken header or first row are not even tried)


def test_hostile_streams_with_true_looking_tables_match_the_oracle(emu, ocodec, datagen, golden):
    rnd = random.Random(5)
    muts, caps, tables = [], [], []
    for size, count in ((150000, 200), (3000, 200)):
        base = ocodec.compress(datagen(size, 60, 9))[1]
        table = make_table(base)
        for t in range(count):
            cc = bytearray(base)
            for _ in range(rnd.randint(1, 3)):
                cc[rnd.randrange(len(cc))] = rnd.randrange(256)
            muts.append(bytes(cc)); caps.append(size); tables.append(table)
    outs, used, rejected = emu_decompress_tables(emu, muts, caps, tables)
    accepted = 0
    for cc, cap, (r, o) in zip(muts, caps, outs):
        ro, oo = ocodec.decompress(cc, cap)
        assert (r < 0) == (ro < 0)
        if r >= 0:
            accepted += 1
            assert r == ro and o == oo
    assert 0 < accepted < len(muts) and used > 0 and rejected > 0


def test_tables_with_history_before_the_block(emu, oracle, datagen):
    d = datagen(400000, 60, 3)
    hist, body = d[:70000], d[70000:]
    # the compressor with history: blocks that reach into the 64 KB before them
    n = 1
    src = ctypes.create_string_buffer(d, len(d))
    cap = len(body) + len(body) // 255 + 16
    dst = ctypes.create_string_buffer(cap + 64)
    stride = hint_bytes(len(body))
    hraw = ctypes.create_string_buffer(stride + 16); hbase = (ctypes.addressof(hraw) + 15) & ~15
    sp = (ctypes.c_void_p * 1)(ctypes.addressof(src) + len(hist)); dp = (ctypes.c_void_p * 1)(ctypes.addressof(dst))
    ss = (ctypes.c_int32 * 1)(len(body)); dc = (ctypes.c_int32 * 1)(cap); res = (ctypes.c_int32 * 1)(); pre = (ctypes.c_int32 * 1)(65536)
    emu.emu_compress_batch_hints(sp, ss, dp, dc, res, 1, 1, pre, ctypes.c_void_p(hbase), ctypes.c_uint64(stride), 1)
    comp = dst.raw[:res[0]]
    table = hraw.raw[hbase - ctypes.addressof(hraw):][:stride]
    check_table(comp, table, len(body))
    outs, used, rejected = emu_decompress_tables(emu, [comp], [len(body)], [table], prefixes=[hist[-65536:]])
    assert outs[0] == (len(body), body) and used == 1 and rejected == 0
    # the same block without its history: offsets reach before the output, with or without the table
    outs, used, rejected = emu_decompress_tables(emu, [comp], [len(body)], [table])
    assert outs[0][0] < 0


def test_a_table_without_room_for_its_rows_is_left_invalid(emu, ocodec, datagen):
    """lz4amd_hint_bytes leaves room for a row per 128 bytes of source (a row is 8 sequences).  A block with more rows than its table
    has room for - here: a stride an eighth of that - gets no table: the header stays invalid, nothing is written behind the
    table's end, and the decoder decodes the block the ordinary way."""
    d = datagen(1 << 20, 90, 2)
    n = 1
    cap = len(d) + len(d) // 255 + 16
    src = ctypes.create_string_buffer(d, len(d)); dst = ctypes.create_string_buffer(cap + 64)
    stride = (hint_bytes(len(d)) // 8) & ~15
    hraw = ctypes.create_string_buffer(b"\xEE" * (stride + 4096 + 16), stride + 4096 + 16)
    hbase = (ctypes.addressof(hraw) + 15) & ~15
    sp = (ctypes.c_void_p * 1)(ctypes.addressof(src)); dp = (ctypes.c_void_p * 1)(ctypes.addressof(dst))
    ss = (ctypes.c_int32 * 1)(len(d)); dc = (ctypes.c_int32 * 1)(cap); res = (ctypes.c_int32 * 1)()
    emu.emu_compress_batch_hints(sp, ss, dp, dc, res, 1, 1, None, ctypes.c_void_p(hbase), ctypes.c_uint64(stride), 1)
    comp = dst.raw[:res[0]]
    ro, o = ocodec.decompress(comp, len(d))
    assert ro == len(d) and o == d
    ch, _ = token_chain(comp)
    assert table_bytes((len(ch) + EVERY - 1) // EVERY) > stride            # (the premise: more rows than room, even at 16 sequences a row)
    off = hbase - ctypes.addressof(hraw)
    table = hraw.raw[off:off + stride]
    assert not is_valid(table)
    assert hraw.raw[off + stride:off + stride + 4096] == b"\xEE" * 4096   # nothing behind the table's end
    outs, used, rejected = emu_decompress_tables(emu, [comp], [len(d)], [table])
    assert outs[0] == (len(d), d) and (used, rejected) == (0, 0)


def test_tables_made_while_decoding_foreign_blocks(emu, foreign):
    """lz4amd_plan_make_hints: the first decode of blocks nobody made a table for writes their tables (stage A knows every token
    then); they hold to the same rules as the compressor's, and the second decode parses from them"""
    blocks = [c for _, c in foreign]
    wants = [d for d, _ in foreign]
    empty = [bytes(hint_bytes(len(d))) for d in wants]
    made = []
    outs, used, rejected = emu_decompress_tables(emu, blocks, [len(d) for d in wants], empty, make=made)
    for d, (r, o) in zip(wants, outs):
        assert r == len(d) and o == d, len(d)
    assert used == 0 and rejected == 0 and made[1] >= len(blocks) - 4, (used, rejected, made[1])
    good = 0
    for d, c, t in zip(wants, blocks, made[0]):
        if is_valid(t):
            check_table(c, t, len(d)); good += 1
        else:
            assert t == bytes(len(t)) or struct.unpack_from("<I", t, 0)[0] == 0        # (left invalid: more rows than room)
    assert good >= len(blocks) - 6, good
    outs, used, rejected = emu_decompress_tables(emu, blocks, [len(d) for d in wants], made[0], salign=3)
    for d, (r, o) in zip(wants, outs):
        assert r == len(d) and o == d, len(d)
    assert used == good and rejected == 0, (used, rejected, good)
    # a table that lies is replaced by a true one on the way
    liars = []
    for t in made[0]:
        b = bytearray(t)
        if is_valid(t) and header(t)[4] > 2:
            tok1, out1, olo1 = unpack_row(t, 1)
            set_row(b, 1, tok1, out1 + 1, olo1)                                          # row 1: output position off by one
        liars.append(bytes(b))
    made2 = []
    outs, used, rejected = emu_decompress_tables(emu, blocks, [len(d) for d in wants], liars, make=made2)
    for d, (r, o) in zip(wants, outs):
        assert r == len(d) and o == d, len(d)
    assert rejected > 0 and made2[1] >= rejected
    assert [t for t in made2[0]] == [t for t in made[0]]


def test_small_blocks_of_compressible_data_get_their_tables(emu, ocodec, datagen):
    """A block's first tile is parsed against an empty table and has no sequences: the row distance of a tile comes from the
    tile's OWN number of sequences (from the tile before it, the second tile got a row per sequence and blocks of a few KB ran
    out of room), and data of fewer than 32 bytes per sequence gets a row per 16 sequences."""
    datas = []
    for pct, seed in ((60, 7), (90, 8), (95, 9)):
        src = datagen(1 << 20, pct, seed)
        for o in range(0, 900000, 151111):
            for n in (1500, 2048, 3000, 4095, 8000, 20000):
                datas.append(src[o:o + n])
    comps, tables = emu_compress_tables(emu, datas)
    for d, c, t in zip(datas, comps, tables):
        check_table(c, t, len(d))
    outs, used, rejected = emu_decompress_tables(emu, comps, [len(d) for d in datas], tables)
    assert used == len(datas) and rejected == 0
    for d, (r, o) in zip(datas, outs):
        assert r == len(d) and o == d


def emu_compress_hc_tables(emu, datas, level):
    n = len(datas)
    caps = [len(d) + len(d) // 255 + 16 for d in datas]
    srcs = [ctypes.create_string_buffer(d, len(d)) if d else ctypes.create_string_buffer(1) for d in datas]
    dsts = [ctypes.create_string_buffer(c + 64) for c in caps]
    stride = max(hint_bytes(len(d)) for d in datas)
    hraw = ctypes.create_string_buffer(stride * n + 16)
    hbase = (ctypes.addressof(hraw) + 15) & ~15
    sp = (ctypes.c_void_p * n)(*[ctypes.addressof(s) for s in srcs])
    dp = (ctypes.c_void_p * n)(*[ctypes.addressof(d) for d in dsts])
    ss = (ctypes.c_int32 * n)(*[len(d) for d in datas]); dc = (ctypes.c_int32 * n)(*caps); res = (ctypes.c_int32 * n)()
    emu.emu_compress_hc_batch_hints(sp, ss, dp, dc, res, n, 0, level, None, ctypes.c_void_p(hbase), ctypes.c_uint64(stride))
    off = hbase - ctypes.addressof(hraw)
    return [dsts[i].raw[:res[i]] for i in range(n)], [hraw.raw[off + i * stride: off + (i + 1) * stride] for i in range(n)]


@pytest.mark.parametrize("level", [9, 2, 12])
def test_hc_compressor_tables_name_real_sequences_and_are_used(emu, ocodec, datagen, level):
    """lz4amd_k_compress_hc writes the same entry-point tables as the fast compressor (its emit lanes, one row per 2^k sequences, k
    per block): every row is a sequence of the block's real token chain, the blocks are byte for byte what they are without the
    table, and the decoder parses all of them from their tables."""
    datas = [datagen(200000, 60, 2), datagen(262144, 60, 3), datagen(300000, 90, 4), datagen(50000, 0, 5), datagen(100, 50, 1), datagen(13, 50, 0), b"z" * 11,
             datagen(5000, 20, 1), datagen(131073, 60, 1), b"\x00" * 300000, b"abcd" * 70000, os.urandom(70000), b"a" * 40000 + os.urandom(3000) + b"a" * 40000]
    if level != 9:
        datas = datas[:6] + datas[9:11]
    comps, tables = emu_compress_hc_tables(emu, datas, level)
    plain = tk.emu_compress_hc(emu, datas, level)
    for d, c, t, (pr, pc) in zip(datas, comps, tables, plain):
        assert c == pc and pr == len(c)
        ro, o = ocodec.decompress(c, len(d))
        assert ro == len(d) and o == d
        check_table(c, t, len(d))
    outs, used, rejected = emu_decompress_tables(emu, comps, [len(d) for d in datas], tables, salign=5)
    for d, (r, o) in zip(datas, outs):
        assert r == len(d) and o == d
    assert used == len(datas) and rejected == 0
