"""the reference's command line program (programs/*.c, unmodified) compiled against
include/ and linked against liblz4_amd.so with the lz4-wlib recipe (programs/Makefile:133-144; oracle/Makefile target
_ref/lz4_amd).  Every file operation of tests/test-lz4-basic.sh:15-85 that matters to the codec runs on the device path and
is cross-checked with the pure reference CLI (oracle/_ref/lz4): each decodes what the other wrote, in every block mode,
with and without checksums, multi-threaded, HC, and in the legacy format (lz4io.c:769-915, 1752-1880), plus `-b` (bench.c)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
OURS, THEIRS = os.path.join(REF, "lz4_amd"), os.path.join(REF, "lz4")


def _run(*cmd, **kw):
    return subprocess.run(list(cmd), capture_output=True, timeout=120, **kw)


@pytest.mark.parametrize("flags", [[], ["-B4", "-BD"], ["-B5", "-BI", "-BX"], ["-B7", "--content-size"], ["-9", "-B6"],
                                   ["-l"], ["-T4", "-B5"], ["--no-frame-crc", "-B4"]])
def test_cli_files_cross_decode(flags, datagen, tmp_path):
    if not (os.path.exists(OURS) and os.path.exists(THEIRS)):
        pytest.skip("oracle/_ref/lz4_amd or lz4 not built (needs /root/reference at build time)")
    src = tmp_path / "in.bin"
    src.write_bytes(datagen(5000000, 60, 21))
    a, b = str(tmp_path / "ours.lz4"), str(tmp_path / "theirs.lz4")
    assert _run(OURS, "-f", *flags, str(src), a).returncode == 0
    assert _run(THEIRS, "-f", *flags, str(src), b).returncode == 0
    ra, rb = str(tmp_path / "a.out"), str(tmp_path / "b.out")
    assert _run(THEIRS, "-d", "-f", a, ra).returncode == 0             # the reference decodes what the device path wrote
    assert _run(OURS, "-d", "-f", b, rb).returncode == 0               # ... and the other way round
    data = src.read_bytes()
    assert open(ra, "rb").read() == data and open(rb, "rb").read() == data
    sa, sb = os.path.getsize(a), os.path.getsize(b)
    assert sa <= 1.03 * sb, (flags, sa, sb)                            # ratio window on whole files
    assert _run(OURS, "-t", a).returncode == 0


def test_cli_bench_mode_and_pipes(datagen, tmp_path):
    if not (os.path.exists(OURS) and os.path.exists(THEIRS)):
        pytest.skip("oracle/_ref/lz4_amd or lz4 not built")
    src = tmp_path / "in.bin"
    src.write_bytes(datagen(3000000, 60, 22))
    r = _run(OURS, "-b1", "-i1", str(src))                             # bench.c: compress + decompress + checksum of the round trip
    assert r.returncode == 0 and b"MB/s" in r.stderr + r.stdout, r.stderr[-300:]
    r = _run(OURS, "-b9", "-e9", "-i1", "-B5", str(src))
    assert r.returncode == 0, r.stderr[-300:]
    c = subprocess.run([OURS, "-c"], input=src.read_bytes(), capture_output=True, timeout=120)
    assert c.returncode == 0
    d = subprocess.run([THEIRS, "-dc"], input=c.stdout, capture_output=True, timeout=120)
    assert d.returncode == 0 and d.stdout == src.read_bytes()


def test_file_helper_example_round_trip(datagen, tmp_path):
    """SURVEY 8(f) item 4, lib/lz4file.h: the reference's examples/fileCompress.c (unmodified; LZ4F_writeOpen / LZ4F_write /
    LZ4F_writeClose, LZ4F_readOpen / LZ4F_read / LZ4F_readClose) compiled against include/lz4file.h and linked against the
    library: compresses a file, decodes it again, verifies; the reference CLI decodes the .lz4 it wrote as well."""
    exe = os.path.join(REF, "fileCompress_amd")
    if not (os.path.exists(exe) and os.path.exists(THEIRS)):
        pytest.skip("oracle/_ref/fileCompress_amd not built (needs /root/reference at build time)")
    src = tmp_path / "in.bin"
    src.write_bytes(datagen(3000000, 60, 23))
    r = _run(exe, str(src))
    assert r.returncode == 0 and b"verify : OK" in r.stdout, (r.stdout[-300:], r.stderr[-300:])
    assert open(str(src) + ".lz4.dec", "rb").read() == src.read_bytes()
    out = str(tmp_path / "ref.out")
    assert _run(THEIRS, "-d", "-f", str(src) + ".lz4", out).returncode == 0
    assert open(out, "rb").read() == src.read_bytes()
