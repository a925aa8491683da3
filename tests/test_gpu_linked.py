"""Linked blocks decoded side by side (lz4_amd/csrc/kernels/chain_spec_kernel.h) against the chain of copy stages they replace
(LZ4AMD_CHAIN_SERIAL=1) and against the source: lz4frame.c:1901-1915 - a block's matches reach 64 KB back into the output before it.
The frames are written by this library's LZ4F_compressFrame (linked blocks, the format's default) and by the reference's CLI
(tests/golden); the blocks are taken out of the frames and handed to lz4amd_plan_create_decompress_chained as device pointers."""
import os
import random

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from test_gpu_parity import ctx  # noqa: E402,F401
from test_gpu_frame import L, compress_frame  # noqa: E402,F401
from test_kernels_emulated import _frame_blocks  # noqa: E402


LAST_STATS = {}


def run_chain(ctx, blocks, cap, total, serial, history=b"", drop_stored=False, twins=None, tables=None):
    """-> (results, output bytes).  blocks: [(stored?, payload)].  LAST_STATS: lz4amd_plan_chain_stats of the last side-by-side plan"""
    import lz4_amd
    payloads = [p for _, p in blocks]
    blob = torch.frombuffer(bytearray(b"".join(payloads) + b"\0" * 64), dtype=torch.uint8).cuda()
    offs = [sum(len(p) for p in payloads[:i]) for i in range(len(payloads))]
    out = torch.full((len(history) + total + cap + 128,), 0xEE, dtype=torch.uint8, device="cuda")
    if history:
        out[16:16 + len(history)] = torch.frombuffer(bytearray(history), dtype=torch.uint8).cuda()
    at = 16 + len(history)
    old = os.environ.pop("LZ4AMD_CHAIN_SERIAL", None)
    if serial:
        os.environ["LZ4AMD_CHAIN_SERIAL"] = "1"
    if twins is not None:                                                    # (the plan's own choice otherwise: lz4amd_batch.c)
        os.environ["LZ4AMD_CHAIN_TWINS"] = "1" if twins else "0"
    if tables is not None:                                                   # (chains of blocks of 1 MiB and more: second copies by a second launch, from tables)
        os.environ["LZ4AMD_CHAIN_TABLES"] = "1" if tables else "0"
    try:
        plan = lz4_amd.Plan.chained(ctx, [blob.data_ptr() + o for o in offs], [len(p) for p in payloads], out.data_ptr() + at,
                                    [cap] * len(payloads), stored=None if drop_stored else [r for r, _ in blocks], initial_prefix=len(history))
    finally:
        os.environ.pop("LZ4AMD_CHAIN_SERIAL", None)
        os.environ.pop("LZ4AMD_CHAIN_TWINS", None)
        os.environ.pop("LZ4AMD_CHAIN_TABLES", None)
        if old is not None:
            os.environ["LZ4AMD_CHAIN_SERIAL"] = old
    res = None
    for _ in range(2):                                                       # a plan is launched again and again
        plan.launch(torch.cuda.current_stream().cuda_stream)
        r = plan.results(torch.cuda.current_stream().cuda_stream)
        assert res is None or r == res
        res = r
    if not serial:
        LAST_STATS.clear()
        try: LAST_STATS.update(plan.chain_stats())
        except RuntimeError: pass                                            # (a chain of one unit decodes block after block)
    plan.close()
    good = sum(r for r in res if r > 0) if all(r >= 0 for r in res) else sum(res[:[i for i, r in enumerate(res) if r < 0][0]])
    host = out.cpu().numpy().tobytes()
    assert host[:16] == b"\xEE" * 16
    return res, host[at:at + good]


def both(ctx, blocks, cap, data, history=b""):
    """the serial chain; side by side with twins (a block's two copies decoded by one workgroup from one record table), with the second copies
    decoded by a second launch from the tables the first ones' decode wrote (blocks of 1 MiB and more; ignored otherwise), and with neither"""
    rs, os_ = run_chain(ctx, blocks, cap, len(data), True, history)
    rt, ot = run_chain(ctx, blocks, cap, len(data), False, history, twins=True, tables=False)
    rb, ob = run_chain(ctx, blocks, cap, len(data), False, history, twins=False, tables=True)
    rp, op = run_chain(ctx, blocks, cap, len(data), False, history, twins=False, tables=False)
    assert rs == rp == rt == rb
    assert os_ == data and op == data and ot == data and ob == data
    return rp


def linked_blocks(L, data, bsid, level=0):
    frame = compress_frame(L, data, level=level, blockSizeID=bsid)
    indep, blocks = _frame_blocks(frame)
    assert not indep
    return blocks


def test_side_by_side_equals_the_serial_chain(ctx, L, datagen):
    rng = random.Random(5)
    noise = rng.randbytes((4 << 20) + 100000)
    cases = [
        (datagen(3 << 20, 60, 1), 4),                                          # 64 KB blocks: every block's history is the whole block before
        (datagen(5 << 20, 50, 2), 5),                                          # 256 KB
        (datagen((9 << 20) + 12345, 60, 3), 7),                                # 4 MB blocks, a short last one
        (datagen(2 << 20, 95, 4), 6),                                          # long matches
        (b"a" * (1 << 20) + b"abcdefg" * 100000 + bytes(range(256)) * 4000, 4),      # every byte a copy of a byte of the first block, period 1 / 7 / 256
        (datagen(200000, 60, 6) + rng.randbytes(150000) + datagen(300000, 60, 7) + rng.randbytes(70000) + datagen(100000, 50, 8), 4),   # stored blocks in the chain
        (datagen(70000, 60, 9), 4),                                            # two blocks
        # a block that begins with a copy from the far end of its 64 KB (offsets above 65280 in a unit's first 255 bytes: the one case that needs
        # the third made-up history), 64 KB blocks and 4 MB blocks
        (noise[:65536] + noise[200:1200] + noise[70000:90000] + noise[66000:66300] * 50, 4),
        (noise[:4 << 20] + noise[(4 << 20) - 65535 + 3:(4 << 20) - 65000] * 3 + datagen(1 << 20, 60, 12), 7),
    ]
    thirds = []
    for data, bsid in cases:
        blocks = linked_blocks(L, data, bsid)
        cap = {4: 65536, 5: 262144, 6: 1 << 20, 7: 4 << 20}[bsid]
        res = both(ctx, blocks, cap, data)
        assert all(r > 0 for r in res) and sum(res) == len(data)
        thirds.append(LAST_STATS.get("units_decoded_three_times", 0))
    # the third decode is the exception (gated: lz4amd_dec_params.chain; chains that do need it: test_gated_third_decode_of_ragged_chains)
    assert thirds[0] == 0 and thirds[1] == 0, thirds
    # HC-compressed linked blocks (matches just behind the write position, and far ones)
    data = datagen(1 << 20, 70, 11)
    both(ctx, linked_blocks(L, data, 4, level=9), 65536, data)


def test_side_by_side_with_history_in_front_and_batches_that_start_anywhere(ctx, L, datagen):
    data = datagen(1 << 20, 60, 21)
    blocks = linked_blocks(L, data, 4)
    for first, hist in ((3, 65536), (5, 65536), (1, 65536)):
        lo = first * 65536
        h = data[lo - hist:lo]
        res = both(ctx, blocks[first:first + 6], 65536, data[lo:lo + 6 * 65536], history=h)
        assert res == [65536] * 6
    # less history than the blocks refer to: the chain fails at its first block, both ways (lz4.c:2356)
    for serial, tw in ((True, None), (False, True), (False, False)):
        res, _ = run_chain(ctx, blocks[3:6], 65536, 3 * 65536, serial, history=data[3 * 65536 - 100:3 * 65536], twins=tw)
        assert all(r < 0 for r in res), (serial, tw, res)
        res, _ = run_chain(ctx, blocks[3:6], 65536, 3 * 65536, serial, twins=tw)
        assert all(r < 0 for r in res), (serial, tw, res)


def test_side_by_side_small_blocks_and_errors(ctx, L, datagen):
    # blocks much smaller than the window: a block's history spans many blocks before it
    import lz4_amd
    data = datagen(300000, 70, 31)
    bs = 3000
    srcs = [data[i:i + bs] for i in range(0, len(data), bs)]
    # compress every piece with the data before it as history (device compressor, linked: lz4amd_plan_create_compress_prefix)
    whole = torch.frombuffer(bytearray(data + b"\0" * 64), dtype=torch.uint8).cuda()
    bound = lz4_amd.compress_bound(bs)
    dst = torch.zeros(len(srcs) * bound + 64, dtype=torch.uint8, device="cuda")
    table = lz4_amd.BlockTable([whole.data_ptr() + i * bs for i in range(len(srcs))], [len(s) for s in srcs],
                               [dst.data_ptr() + i * bound for i in range(len(srcs))], [bound] * len(srcs))
    plan = lz4_amd.Plan.compress_with_history(ctx, table, [min(i * bs, 65536) for i in range(len(srcs))])
    plan.launch(torch.cuda.current_stream().cuda_stream)
    cs = plan.results(torch.cuda.current_stream().cuda_stream)
    plan.close()
    host = dst.cpu().numpy().tobytes()
    blocks = [(False, host[i * bound:i * bound + c]) for i, c in enumerate(cs)]
    assert sum(cs) < len(data) * 0.6                                          # (the history is used)
    res = both(ctx, blocks, bs, data)
    assert res == [len(s) for s in srcs]
    # a damaged block in the middle ends the chain there, the blocks before it stand
    bad = list(blocks); bad[40] = (False, bad[40][1][:-5])
    for serial, tw in ((True, None), (False, True), (False, False)):
        res, out = run_chain(ctx, bad, bs, len(data), serial, twins=tw)
        assert res[:40] == [bs] * 40 and all(r < 0 for r in res[40:]), (serial, tw)
        assert out == data[:40 * bs]


def _len_bytes(n):
    out = b""
    while n >= 255: out += b"\xff"; n -= 255
    return out + bytes([n])


def _block(lits0, match, lits1):
    """A hand-made LZ4 block: lits0, one match (offset, length) or None, lits1 as the final literals (>= 12 bytes behind a match: lz4.c:2170)"""
    out = b""
    if match:
        off, ml = match
        out += bytes([(min(len(lits0), 15) << 4) | min(ml - 4, 15)]) + (_len_bytes(len(lits0) - 15) if len(lits0) >= 15 else b"") + lits0
        out += bytes([off & 255, off >> 8]) + (_len_bytes(ml - 4 - 15) if ml - 4 >= 15 else b"")
    else:
        lits1 = lits0 + lits1
    return out + bytes([min(len(lits1), 15) << 4]) + (_len_bytes(len(lits1) - 15) if len(lits1) >= 15 else b"") + lits1


def test_gated_third_decode_of_ragged_chains(ctx):
    """Chains of blocks of any size (hand-made blocks): a unit of the side-by-side decode is as many blocks as make 1 MiB of capacity.  A unit's
    third decode is gated on what the first block of its first decode reports - unless that block is so small that a later one could begin
    within the unit's first 255 bytes and read the far end of the 64 KB itself: then nobody is asked."""
    rng = random.Random(77)
    cap = 65536                                                              # -> units of 16 blocks
    noise = rng.randbytes(16 * cap)
    S = 16 * cap
    for head, far_at, want_third in ((100, 100, 1), (600, 10, 1), (600, None, 0), (100, None, 0), (600, 700, 0)):
        # unit 1 starts at S with a block of `head` bytes; 40 bytes at unit position far_at stand 65526 bytes back: in the unit's first 255 bytes
        # they are a copy of one of the first 256 bytes of the unit's 64 KB (in the second block when head = 100), at 700 they are not
        tail = rng.randbytes(3 * cap - 1000)
        blocks = [(False, _block(noise[i * cap:(i + 1) * cap], None, b"")) for i in range(16)]
        data = bytearray(noise)
        hb = rng.randbytes(head)
        if far_at is not None and far_at < head:
            pre, post = hb[:far_at], hb[far_at + 40:]
            blocks.append((False, _block(pre, (65526, 40), post))); data += pre; data += data[len(data) - 65526:len(data) - 65526 + 40]; data += post
        else:
            blocks.append((False, _block(hb, None, b""))); data += hb
        if far_at is not None and far_at >= head:
            pre = rng.randbytes(far_at - head)
            blocks.append((False, _block(pre, (65526, 40), tail[:cap - len(pre) - 40]))); data += pre; data += data[len(data) - 65526:len(data) - 65526 + 40]; data += tail[:cap - len(pre) - 40]
        else:
            blocks.append((False, _block(tail[:cap], None, b""))); data += tail[:cap]
        # a block that copies from the blocks before it (the unit's bytes depend on the history all along), and a last short one
        blocks.append((False, _block(b"abc", (30000, 50000), tail[:1000]))); data += b"abc"
        for _ in range(50000): data.append(data[len(data) - 30000])
        data += tail[:1000]
        blocks.append((False, _block(b"", (65535, 20), b"the end of it")));
        for _ in range(20): data.append(data[len(data) - 65535])
        data += b"the end of it"
        data = bytes(data)
        res = both(ctx, blocks, cap, data)
        assert sum(res) == len(data) and all(r > 0 for r in res)
        assert LAST_STATS["units"] == 2 and LAST_STATS["units_decoded_three_times"] == want_third, (head, far_at, LAST_STATS)


def test_large_blocks_second_launch_from_tables(ctx, L, datagen):
    """Chains of blocks of 1 MiB and more: the second copies (and the third, where a unit needs it) are decoded by a second launch from the tables
    the first copies' decode wrote (spec_gate).  Hand-made and compressed chains, a damaged block, a unit that needs its third copy."""
    rng = random.Random(78)
    cap = 1 << 20
    noise = rng.randbytes(cap)
    for far_at, want_third in ((10, 1), (300, 0)):
        body = datagen(600000, 60, 5 + far_at)
        pre = rng.randbytes(far_at)
        blocks = [(False, _block(noise, None, b"")), (False, _block(pre, (65530, 40), body))]
        data = bytearray(noise + pre)
        for _ in range(40): data.append(data[len(data) - 65530])
        data += body
        blocks.append((False, _block(b"xyz", (70, 3000), b"and so it ends")))
        data += b"xyz"
        for _ in range(3000): data.append(data[len(data) - 70])
        data += b"and so it ends"
        res = both(ctx, blocks, cap, bytes(data))
        assert sum(res) == len(data)
        assert LAST_STATS["units"] == 3 and LAST_STATS["units_decoded_three_times"] == want_third, (far_at, LAST_STATS)
    data = datagen(5 << 20, 60, 41)
    blocks = linked_blocks(L, data, 6)
    both(ctx, blocks, cap, data)
    bad = list(blocks); bad[2] = (False, bad[2][1][:-9])
    for tb in (True, False):
        res, out = run_chain(ctx, bad, cap, len(data), False, tables=tb)
        assert res[:2] == [cap, cap] and all(r < 0 for r in res[2:]), tb
        assert out == data[:2 * cap]
