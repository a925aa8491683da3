import ctypes
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure(path, cmd):
    if not os.path.exists(path):
        subprocess.run(cmd, check=True, cwd=ROOT)
    return path


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    """The plain-C CPU restatement (test infrastructure)."""
    so = _ensure(os.path.join(ROOT, "oracle", "liblz4oracle.so"), ["make", "-C", "oracle", "liblz4oracle.so"])
    L = ctypes.CDLL(so)
    L.lz4o_xxh32.restype = ctypes.c_uint32
    L.lz4o_xxh32.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32]
    L.lz4o_frame_compress.restype = ctypes.c_size_t
    L.lz4o_frame_compress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t] + [ctypes.c_int] * 4
    L.lz4o_frame_decompress.restype = ctypes.c_size_t
    L.lz4o_frame_decompress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.lz4o_frame_bound.restype = ctypes.c_size_t
    L.lz4o_frame_bound.argtypes = [ctypes.c_size_t] + [ctypes.c_int] * 3
    return L


@pytest.fixture(scope="session")
def reflib():
    """The real reference compiled into oracle/_ref (absent on the GPU box unless prebuilt)."""
    so = os.path.join(ROOT, "oracle", "_ref", "liblz4_ref.so")
    if not os.path.exists(so):
        if os.path.isdir("/root/reference/lib"):
            subprocess.run(["make", "-C", "oracle", "ref"], check=True, cwd=ROOT)
        else:
            pytest.skip("oracle/_ref not built and /root/reference absent")
    L = ctypes.CDLL(so)
    L.LZ4_XXH32.restype = ctypes.c_uint32
    return L


@pytest.fixture(scope="session")
def datagen():
    so = _ensure(os.path.join(ROOT, "tools", "libdatagen.so"),
                 ["gcc", "-O3", "-shared", "-fPIC", "-o", "tools/libdatagen.so", "tools/datagen.c"])
    L = ctypes.CDLL(so)
    L.lz4amd_datagen.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, ctypes.c_uint32]

    def gen(n, pct, seed=0):
        buf = ctypes.create_string_buffer(max(n, 1))
        assert L.lz4amd_datagen(buf, n, pct / 100.0, 0.0, seed) == 0
        return buf.raw[:n]
    return gen


@pytest.fixture(scope="session")
def emu():
    """CPU-interpreted twins of the kernels (tests/simt) -- logic tests without a GPU."""
    so = os.path.join(ROOT, "tests", "simt", "libemu_kernels.so")
    srcs = [os.path.join(ROOT, "tests", "simt", f) for f in ("emu_kernels.cpp", "simt_emu.cpp", "simt_emu.h", "platform_emu.h")]
    kdir = os.path.join(ROOT, "lz4_amd", "csrc", "kernels")
    srcs += [os.path.join(kdir, f) for f in os.listdir(kdir)]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["sh", os.path.join(ROOT, "tests", "simt", "build.sh")], check=True)
    return ctypes.CDLL(so)


def md5(b):
    return hashlib.md5(b).hexdigest()


class OracleCodec:
    """Convenience wrappers over the oracle for the tests."""

    def __init__(self, L):
        self.L = L

    def bound(self, n):
        return self.L.lz4o_compress_bound(n)

    def compress(self, data, cap=None, accel=1):
        cap = self.bound(len(data)) if cap is None else cap
        dst = ctypes.create_string_buffer(max(cap, 1))
        r = self.L.lz4o_compress_fast(data, dst, len(data), cap, accel)
        return r, dst.raw[:max(r, 0)]

    def decompress(self, comp, cap):
        dst = ctypes.create_string_buffer(max(cap, 1) + 8)
        r = self.L.lz4o_decompress_safe(comp, dst, len(comp), cap)
        return r, dst.raw[:max(r, 0)]


@pytest.fixture(scope="session")
def ocodec(oracle):
    return OracleCodec(oracle)
