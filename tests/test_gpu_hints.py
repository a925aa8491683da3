"""Entry-point tables (include/lz4amd.h) on the MI355X: lz4amd_k_compress's tables name real sequences, the decoder's
PARSER path decodes our own blocks and - with tables made here - blocks of any origin bit-exactly, tables that lie only
cost time, and hostile streams behind true-looking tables are judged exactly as the oracle judges them.
Helpers shared with the CPU-interpreter twin of this file (tests/test_hints_emulated.py)."""
import os
import random
import struct

import pytest

import test_hints_emulated as th
import test_kernels_emulated as tk

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    import lz4_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return lz4_amd.Context(0)


def _dev(b, pad=0, fill=0xEE):
    t = torch.full((len(b) + pad,), fill, dtype=torch.uint8, device="cuda")
    if len(b):
        t[:len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
    return t


def gpu_compress_tables(ctx, datas, acceleration=1, hc_level=None):
    import lz4_amd
    s = torch.cuda.current_stream().cuda_stream
    caps = [lz4_amd.compress_bound(len(d)) for d in datas]
    srcs = [_dev(d, pad=16) for d in datas]
    dsts = [torch.full((c + 64,), 0xEE, dtype=torch.uint8, device="cuda") for c in caps]
    stride = max(lz4_amd.hint_bytes(len(d)) for d in datas)
    hints = torch.full((len(datas), stride), 0xEE, dtype=torch.uint8, device="cuda")
    tab = lz4_amd.BlockTable([x.data_ptr() for x in srcs], [len(d) for d in datas], [x.data_ptr() for x in dsts], caps)
    plan = lz4_amd.Plan(ctx, lz4_amd.OP_COMPRESS, tab) if hc_level is None else lz4_amd.Plan(ctx, lz4_amd.OP_COMPRESS_HC, tab, level=hc_level)
    plan.attach_hints(hints.data_ptr(), stride)
    if acceleration != 1:
        plan.set_acceleration(acceleration)
    plan.launch(s)
    res = plan.results(s)
    h = hints.cpu().numpy()
    comps = []
    for r, d, cap in zip(res, dsts, caps):
        raw = d.cpu().numpy().tobytes()
        assert raw[cap:] == b"\xEE" * 64
        comps.append(raw[:max(r, 0)])
    return comps, [h[i].tobytes() for i in range(len(datas))]


def gpu_decompress_tables(ctx, blocks, caps, tables, salign=0, guard=64, make=None):
    import lz4_amd
    s = torch.cuda.current_stream().cuda_stream
    srcs = [_dev(b"\xA5" * salign + c, pad=16) for c in blocks]
    dsts = [torch.full((max(c, 0) + guard,), 0xEE, dtype=torch.uint8, device="cuda") for c in caps]
    stride = max(16 * ((len(t) + 15) // 16) for t in tables)
    hints = torch.zeros((len(blocks), stride), dtype=torch.uint8, device="cuda")
    for i, t in enumerate(tables):
        hints[i, :len(t)] = torch.frombuffer(bytearray(t), dtype=torch.uint8).cuda()
    plan = lz4_amd.Plan(ctx, lz4_amd.OP_DECOMPRESS, lz4_amd.BlockTable([x.data_ptr() + salign for x in srcs], [len(c) for c in blocks],
                                                                      [x.data_ptr() for x in dsts], caps))
    plan.attach_hints(hints.data_ptr(), stride)
    if make is not None:
        plan.make_hints(True)
    plan.launch(s)
    res = plan.results(s)
    used, rejected = plan.hint_stats()
    if make is not None:
        h = hints.cpu().numpy()
        make[:] = [[h[i].tobytes() for i in range(len(blocks))], plan.hints_made()]
    outs = []
    for r, d, cap in zip(res, dsts, caps):
        raw = d.cpu().numpy().tobytes()
        assert raw[max(cap, 0):] == b"\xEE" * guard, "wrote past dst[cap]"
        outs.append((r, raw[:max(r, 0)]))
    return outs, used, rejected


def test_compressor_tables_name_real_sequences_and_are_used(ctx, ocodec, datagen):
    import lz4_amd
    specs = [(200000, 60, 2), (65536, 50, 0), (1 << 20, 60, 3), (300000, 90, 4), (50000, 0, 5), (100, 50, 1), (13, 50, 0), (5000, 20, 1),
             (131073, 60, 1), (4 << 20, 60, 0), (4 << 20, 20, 1), (4 << 20, 90, 2), (1024, 60, 1), (1025, 60, 1), (65535, 60, 1)]
    datas = [datagen(*s) for s in specs] + [b"\x00" * 300000, b"abcd" * 70000, os.urandom(70000), b"a" * 40000 + os.urandom(3000) + b"a" * 40000,
                                            os.urandom(1 << 20) + b"q" * 100000 + os.urandom(50000)] + tk._region_index_corpus()[2:5]
    comps, tables = gpu_compress_tables(ctx, datas)
    for d, c, t in zip(datas, comps, tables):
        ro, o = ocodec.decompress(c, len(d))
        assert ro == len(d) and o == d
        th.check_table(c, t, len(d))
    for sal in (0, 9):
        outs, used, rejected = gpu_decompress_tables(ctx, comps, [len(d) for d in datas], tables, salign=sal)
        for d, (r, o) in zip(datas, outs):
            assert r == len(d) and o == d
        assert used == len(datas) and rejected == 0
    outs, used, _ = gpu_decompress_tables(ctx, comps[:3], [len(d) - 1 for d in datas[:3]], tables[:3])
    assert all(r < 0 for r, _ in outs) and used == 0


def test_tables_made_for_foreign_blocks_decode_every_corpus(ctx, ocodec, reflib, datagen):
    foreign = th.foreign_cases(ocodec, reflib, datagen)
    blocks = [c for _, c in foreign]
    wants = [d for d, _ in foreign]
    for every, by_bytes, sal in ((8, 0, 0), (8, 0, 5), (1, 0, 0), (0, 512, 0), (0, 3000, 3)):
        tables = [th.make_table(c, every, by_bytes) for c in blocks]
        outs, used, rejected = gpu_decompress_tables(ctx, blocks, [len(d) for d in wants], tables, salign=sal)
        for d, (r, o) in zip(wants, outs):
            assert r == len(d) and o == d, (every, by_bytes, sal, len(d))
        # (rows every 3000 bytes of a block of 6-byte sequences are more than 255 sequences apart: the rows' 8 bits cannot say that, such a table
        #  is rejected - not wrong; tests/test_hints_emulated.py has the case with rows 2000 sequences apart as well)
        assert used + rejected == len(blocks) and (rejected == 0 or by_bytes == 3000) and used >= len(blocks) - 2, (every, by_bytes, used, rejected)


def test_tables_that_lie_only_cost_time(ctx, ocodec, datagen):
    rnd = random.Random(77)
    d = datagen(300000, 60, 5)
    c = ocodec.compress(d)[1]
    good = th.make_table(c)
    nrows = th.header(good)[4]
    blocks, tables = [], []
    for t in range(48):
        bad = bytearray(good)
        kind = t % 4
        if kind == 0:
            for _ in range(rnd.randint(1, 4)):
                struct.pack_into("<I", bad, 32 + 4 * rnd.randrange(2 * (nrows - 1)), rnd.randrange(1 << 22))
        elif kind == 1:
            r = rnd.randrange(1, nrows - 1)
            tok, out, olo = th.unpack_row(bad, r)
            th.set_row(bad, r, tok + rnd.choice((1, 2, 3)), out + rnd.choice((0, 1)), olo)
        elif kind == 2:
            r = rnd.randrange(1, nrows - 1)
            tok, out, olo = th.unpack_row(bad, r)
            th.set_row(bad, r, tok, out + (1 if t % 8 < 4 else 0), olo + (0 if t % 8 < 4 else 1))
        else:
            bad = bytearray(rnd.randbytes(len(good)))
            struct.pack_into("<8I", bad, 0, *struct.unpack_from("<8I", good, 0))
        blocks.append(c); tables.append(bytes(bad))
    outs, used, rejected = gpu_decompress_tables(ctx, blocks, [len(d)] * len(blocks), tables)
    for r, o in outs:
        assert r == len(d) and o == d
    assert used + rejected >= 36 and rejected >= 30               # (a table whose first row is broken is not even tried)


def test_hostile_streams_with_true_looking_tables_match_the_oracle(ctx, ocodec, datagen):
    rnd = random.Random(5)
    muts, caps, tables = [], [], []
    for size, count in ((150000, 250), (3000, 250)):
        base = ocodec.compress(datagen(size, 60, 9))[1]
        table = th.make_table(base)
        for t in range(count):
            cc = bytearray(base)
            for _ in range(rnd.randint(1, 3)):
                cc[rnd.randrange(len(cc))] = rnd.randrange(256)
            muts.append(bytes(cc)); caps.append(size); tables.append(table)
    outs, used, rejected = gpu_decompress_tables(ctx, muts, caps, tables)
    accepted = 0
    for cc, cap, (r, o) in zip(muts, caps, outs):
        ro, oo = ocodec.decompress(cc, cap)
        assert (r < 0) == (ro < 0)
        if r >= 0:
            accepted += 1
            assert r == ro and o == oo
    assert 0 < accepted < len(muts) and used > 0 and rejected > 0


def test_full_size_round_trip_with_tables(ctx, oracle, datagen):
    """BASELINE configs[1]'s block size, 64 blocks: compress with tables, decode from them; every block also through the
    oracle decoder (the CPU restatement of LZ4_decompress_safe)."""
    import ctypes
    import lz4_amd
    bs, nb = 4 << 20, 64
    host = datagen(nb * bs, 60, 3)
    data = torch.frombuffer(bytearray(host), dtype=torch.uint8).cuda()
    hints = torch.zeros((nb, lz4_amd.hint_bytes(bs)), dtype=torch.uint8, device="cuda")
    comp, csizes, _ = lz4_amd.compress_blocks(ctx, data, bs, hints=hints)
    out, res, plan = lz4_amd.decompress_blocks(ctx, comp, csizes, bs, nb * bs, hints=hints)
    assert res == [bs] * nb and torch.equal(out, data)
    assert plan.hint_stats() == (nb, 0)
    hc = comp.cpu().numpy()
    dst = ctypes.create_string_buffer(bs)
    for i in range(nb):
        r = oracle.lz4o_decompress_safe(hc[i, :csizes[i]].tobytes(), dst, csizes[i], bs)
        assert r == bs and dst.raw == host[i * bs:(i + 1) * bs]


def test_tables_made_while_decoding_foreign_blocks(ctx, ocodec, reflib, datagen):
    """lz4amd_plan_make_hints: twin of the interpreter test - plus 4 MiB reference-compressed blocks at three compressibilities"""
    import ctypes
    foreign = th.foreign_cases(ocodec, reflib, datagen)
    for spec in ((4 << 20, 60, 7), (4 << 20, 20, 8), (4 << 20, 90, 9)):
        d = datagen(*spec)
        cap = len(d) + len(d) // 255 + 16
        cb = ctypes.create_string_buffer(cap)
        n = reflib.LZ4_compress_default(d, cb, len(d), cap)
        foreign.append((d, cb.raw[:n]))
    blocks = [c for _, c in foreign]
    wants = [d for d, _ in foreign]
    empty = [bytes(th.hint_bytes(len(d))) for d in wants]
    made = []
    outs, used, rejected = gpu_decompress_tables(ctx, blocks, [len(d) for d in wants], empty, make=made)
    for d, (r, o) in zip(wants, outs):
        assert r == len(d) and o == d, len(d)
    assert used == 0 and rejected == 0 and made[1] >= len(blocks) - 4, (used, rejected, made[1])
    good = 0
    for d, c, t in zip(wants, blocks, made[0]):
        if th.is_valid(t):
            th.check_table(c, t, len(d)); good += 1
    assert good >= len(blocks) - 6, good
    outs, used, rejected = gpu_decompress_tables(ctx, blocks, [len(d) for d in wants], made[0], salign=3)
    for d, (r, o) in zip(wants, outs):
        assert r == len(d) and o == d, len(d)
    assert used == good and rejected == 0, (used, rejected, good)


@pytest.mark.parametrize("level", [9, 2, 12])
def test_hc_compressor_tables_name_real_sequences_and_are_used(ctx, ocodec, datagen, level):
    """lz4amd_k_compress_hc writes the tables as well (its emit lanes; one row distance per block): rows are real sequences, and
    the decoder parses every block from its table - configs[3]'s blocks then decode like the fast compressor's."""
    datas = [datagen(262144, 60, 3), datagen(200000, 60, 2), datagen(300000, 90, 4), datagen(50000, 0, 5), datagen(100, 50, 1), datagen(13, 50, 0), b"z" * 11,
             datagen(5000, 20, 1), datagen(131073, 60, 1), datagen(1 << 20, 60, 7), b"\x00" * 300000, b"abcd" * 70000, os.urandom(70000), b"a" * 40000 + os.urandom(3000) + b"a" * 40000]
    comps, tables = gpu_compress_tables(ctx, datas, hc_level=level)
    for d, c, t in zip(datas, comps, tables):
        ro, o = ocodec.decompress(c, len(d))
        assert ro == len(d) and o == d
        th.check_table(c, t, len(d))
    outs, used, rejected = gpu_decompress_tables(ctx, comps, [len(d) for d in datas], tables, salign=3)
    for d, (r, o) in zip(datas, outs):
        assert r == len(d) and o == d
    assert used == len(datas) and rejected == 0
