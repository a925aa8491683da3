"""The idea behind lz4_amd/csrc/kernels/chain_spec_kernel.h, checked on the CPU with the oracle's decoder (no GPU, no product code):
a byte of LZ4 output is a copy of exactly one earlier byte, so a linked block (lz4frame.c:1901-1915) decoded against made-up
64 KB histories - byte i (lo = i & 0xFF, hi = i >> 8): A = lo, B = lo + (255 - hi) + 1 mod 256, C = ~lo - tells for every output byte whether it
comes from the history (A != B - or, for the first 256 bytes of the 64 KB, where B = A: A != C) and from which byte of it ({255 - (B - A - 1), A});
putting the real history's bytes there gives the block as decoded behind its real predecessors.  Which blocks need C is what the product asks
its decoder (lowref): here it is computed from all three.  The blocks are
those of a linked frame written by the reference's CLI (tests/golden) and of frames the oracle writes."""
import ctypes
import hashlib
import os

import numpy as np

from conftest import GOLDEN_DIR
from test_kernels_emulated import _frame_blocks

HIST = 65536
I = np.arange(HIST, dtype=np.uint32)
MADE_UP = [(I & 0xFF).astype(np.uint8), (((I & 0xFF) + (255 - (I >> 8)) + 1) & 0xFF).astype(np.uint8), (~I & 0xFF).astype(np.uint8)]
NEEDED_C = []


def decode_against(oracle, payload, cap, history):
    buf = ctypes.create_string_buffer(len(history) + cap + 64)
    ctypes.memmove(buf, bytes(history), len(history))
    dst = ctypes.cast(ctypes.addressof(buf) + len(history), ctypes.c_void_p)
    oracle.lz4o_decompress_safe_prefix.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    r = oracle.lz4o_decompress_safe_prefix(payload, dst, len(payload), cap, len(history))
    return r, np.frombuffer(buf.raw[len(history):len(history) + max(r, 0)], dtype=np.uint8)


def side_by_side(oracle, blocks, cap):
    out = np.zeros(0, dtype=np.uint8)
    dependent = []
    for stored, payload in blocks:
        if stored:
            out = np.concatenate([out, np.frombuffer(payload, dtype=np.uint8)]); dependent.append(0)
            continue
        if len(out) == 0:
            r, a = decode_against(oracle, payload, cap, b"")
            assert r >= 0
            out = a.copy(); dependent.append(0)
            continue
        (ra, a), (rb, b), (rc, c) = (decode_against(oracle, payload, cap, h.tobytes()) for h in MADE_UP)
        assert ra == rb == rc and ra >= 0                                     # the sizes do not depend on what the history holds
        dep = a != c                                                          # C differs from A in every byte of the history
        two = a != b                                                          # ... B in all but the first 256
        assert not (two & ~dep).any()
        idx = ((255 - ((b.astype(np.int64) - a.astype(np.int64) - 1) & 0xFF)) << 8) | a
        assert (idx[dep & ~two] < 256).all()                                  # what only C finds are copies of the first 256 bytes of the 64 KB
        NEEDED_C.append(bool((dep & ~two).any()))
        back = HIST - idx[dep]                                                # 1 .. 65536 bytes before the block
        assert (back <= len(out)).all()                                       # (lz4.c:2356 otherwise)
        blk = a.copy()
        blk[dep] = out[len(out) - back]
        dependent.append(int(dep.sum()))
        out = np.concatenate([out, blk])
    return out.tobytes(), dependent


def test_reference_written_linked_frame(oracle, golden):
    g = golden["frames"]["f_p60_600k_B4_BD_cs"]
    frame = open(os.path.join(GOLDEN_DIR, "f_p60_600k_B4_BD_cs.lz4"), "rb").read()
    indep, blocks = _frame_blocks(frame)
    assert not indep
    out, dependent = side_by_side(oracle, blocks, 65536)
    assert hashlib.md5(out).hexdigest() == g["src_md5"]
    assert dependent[0] == 0 and all(d > 0 for d in dependent[1:])             # every block leans on the one before


def test_linked_frames_of_many_kinds(oracle, reflib, datagen):
    """Frames the reference writes here (oracle/_ref: LZ4F_compressFrame, linked blocks are its default): long periods, stored blocks in the chain,
    HC matches."""
    import random
    from test_gpu_frame import Prefs
    rng = random.Random(9)
    st = ctypes.c_size_t
    reflib.LZ4F_compressFrameBound.restype = st
    reflib.LZ4F_compressFrameBound.argtypes = [st, ctypes.POINTER(Prefs)]
    reflib.LZ4F_compressFrame.restype = st
    reflib.LZ4F_compressFrame.argtypes = [ctypes.c_char_p, st, ctypes.c_char_p, st, ctypes.POINTER(Prefs)]
    cases = [(datagen(400000, 60, 1), 4, 0), (datagen(700000, 90, 2), 5, 0), (b"a" * 300000 + b"abcdefg" * 30000 + bytes(range(256)) * 500, 4, 0),
             (datagen(150000, 50, 3) + rng.randbytes(140000) + datagen(150000, 70, 4), 4, 0), (datagen(300000, 70, 5), 4, 9)]
    for data, bsid, level in cases:
        p = Prefs()
        p.frameInfo.blockSizeID = bsid
        p.compressionLevel = level
        cap = reflib.LZ4F_compressFrameBound(len(data), ctypes.byref(p))
        dst = ctypes.create_string_buffer(cap)
        n = reflib.LZ4F_compressFrame(dst, cap, data, len(data), ctypes.byref(p))
        assert 0 < n <= cap
        indep, blocks = _frame_blocks(dst.raw[:n])
        assert not indep and len(blocks) > 1
        out, _ = side_by_side(oracle, blocks, {4: 65536, 5: 262144}[bsid])
        assert out == data
