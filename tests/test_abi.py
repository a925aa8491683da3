"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares; the
device-dependent entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def libpath():
    from lz4_amd import build
    return build.build_product()


def _declared_functions():
    names = set()
    inc = os.path.join(ROOT, "include")
    for h in os.listdir(inc):
        text = open(os.path.join(inc, h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
        for m in re.finditer(r"\b((?:LZ4F?|lz4amd)_[A-Za-z0-9_]+)\s*\(", text):
            names.add(m.group(1))
    return names


def test_exports_every_declared_symbol(libpath):
    out = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    declared = _declared_functions()
    assert declared, "no prototypes parsed"
    missing = sorted(declared - exported)
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_pure_arithmetic_entry_points(libpath):
    L = ctypes.CDLL(libpath)
    assert L.LZ4_compressBound(4 << 20) == 4210768
    assert L.LZ4_compressBound(65536) == 65809
    assert L.LZ4_compressBound(0x7E000001) == 0
    assert L.lz4amd_compress_bound(262144) == 263188
    L.LZ4_versionString.restype = ctypes.c_char_p
    assert L.LZ4_versionNumber() == 11000 and L.LZ4_versionString() == b"1.10.0"
    assert L.LZ4_sizeofState() == 16416


def test_no_silent_cpu_fallback(libpath):
    """Without a device the classic API must report failure, not quietly run on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = ctypes.CDLL(libpath)
    dst = ctypes.create_string_buffer(128)
    assert L.LZ4_compress_default(b"hello hello hello hello", dst, 23, 128) == 0
    assert L.LZ4_decompress_safe(b"\x00", dst, 1, 128) < 0
    h = ctypes.c_void_p()
    assert L.lz4amd_ctx_create(ctypes.byref(h), 0) == -1     # LZ4AMD_E_NODEVICE


def test_product_does_not_link_oracle(libpath):
    out = subprocess.run(["nm", "-D", libpath], capture_output=True, text=True, check=True).stdout
    assert "lz4o_" not in out
    ldd = subprocess.run(["ldd", libpath], capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "liblz4_ref" not in ldd


def test_bounds_match_the_reference(libpath, reflib):
    """Pure arithmetic entry points against the real reference (oracle/_ref): LZ4F_compressBound (lz4frame.c:379-424),
    LZ4F_compressFrameBound (406-416), LZ4_compressBound, LZ4_decoderRingBufferSize (lz4.c:2622-2628)."""
    import itertools

    class FrameInfo(ctypes.Structure):
        _fields_ = [("blockSizeID", ctypes.c_int), ("blockMode", ctypes.c_int), ("contentChecksumFlag", ctypes.c_int),
                    ("frameType", ctypes.c_int), ("contentSize", ctypes.c_ulonglong), ("dictID", ctypes.c_uint),
                    ("blockChecksumFlag", ctypes.c_int)]

    class Prefs(ctypes.Structure):
        _fields_ = [("frameInfo", FrameInfo), ("compressionLevel", ctypes.c_int), ("autoFlush", ctypes.c_uint),
                    ("favorDecSpeed", ctypes.c_uint), ("reserved", ctypes.c_uint * 3)]

    L = ctypes.CDLL(libpath)
    for lib in (L, reflib):
        for name in ("LZ4F_compressBound", "LZ4F_compressFrameBound"):
            getattr(lib, name).restype = ctypes.c_size_t
            getattr(lib, name).argtypes = [ctypes.c_size_t, ctypes.c_void_p]
    sizes = (0, 1, 65535, 65536, 65537, 262144, 1000000, (4 << 20) - 1, 4 << 20, (4 << 20) + 1, 50000000)
    for n in sizes:
        assert L.LZ4F_compressBound(n, None) == reflib.LZ4F_compressBound(n, None), n
        assert L.LZ4F_compressFrameBound(n, None) == reflib.LZ4F_compressFrameBound(n, None), n
    for bsid, bx, cs, af, n in itertools.product((0, 4, 5, 6, 7), (0, 1), (0, 1), (0, 1), sizes):
        p = Prefs()
        p.frameInfo.blockSizeID, p.frameInfo.blockChecksumFlag, p.frameInfo.contentChecksumFlag, p.autoFlush = bsid, bx, cs, af
        assert L.LZ4F_compressBound(n, ctypes.byref(p)) == reflib.LZ4F_compressBound(n, ctypes.byref(p)), (bsid, bx, cs, af, n)
        assert L.LZ4F_compressFrameBound(n, ctypes.byref(p)) == reflib.LZ4F_compressFrameBound(n, ctypes.byref(p)), (bsid, bx, cs, af, n)
    for n in (-1, 0, 1, 15, 16, 17, 65536, 4 << 20, 0x7E000000, 0x7E000001):
        assert L.LZ4_compressBound(n) == reflib.LZ4_compressBound(n)
        assert L.LZ4_decoderRingBufferSize(n) == reflib.LZ4_decoderRingBufferSize(n), n
    assert L.LZ4F_compressionLevel_max() == reflib.LZ4F_compressionLevel_max() == 12
    assert L.LZ4_sizeofStateHC() == reflib.LZ4_sizeofStateHC()
    assert L.LZ4_sizeofState() == reflib.LZ4_sizeofState()


def test_ignored_arguments_leave_a_notice():
    """acceleration > 2 and HC levels > 10 are accepted and not (fully) acted on; the ABI says so (lz4amd_last_notice), also
    without a device: the notice is recorded before the block is touched."""
    import ctypes
    import lz4_amd
    L = lz4_amd.lib()
    L.lz4amd_last_notice.restype = ctypes.c_char_p
    L.LZ4_compress_fast.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.LZ4_compress_HC.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    dst = ctypes.create_string_buffer(64)
    L.LZ4_compress_fast(b"abc", dst, -1, 64, 8)                      # invalid size: returns 0 at once, notice still set
    assert b"acceleration" in L.lz4amd_last_notice()
    for accel in (1, 2):                                               # the two settings the parse knows
        L.LZ4_compress_fast(b"abc", dst, -1, 64, accel)
        assert L.lz4amd_last_notice() == b""
    L.LZ4_compress_HC(b"abc", dst, -1, 64, 12)
    assert b"level 10" in L.lz4amd_last_notice()                       # levels 11-12: their own search depth, level 10's sufficient length
    L.LZ4_compress_HC(b"abc", dst, -1, 64, 10)
    assert L.lz4amd_last_notice() == b""
    L.LZ4_compress_HC(b"abc", dst, -1, 64, 9)
    assert L.lz4amd_last_notice() == b""
