"""The entry-point table's layout (lz4_amd/csrc/lz4amd_params.h), restated for the tests that build, check and falsify tables:
32 bytes of header - { magic, out_size, csize, nseq }, { nrows, 0, 0, 0 } - then nrows + 1 rows of 8 bytes
{ token position (24 bits) | sequences before the row mod 256 (8 bits), output position }; row 0 is { 0, 0 }, row nrows the block's end."""
import struct

MAGIC = 0x32485A4C
HEAD = 32
ROW = 8


def hint_bytes(n):
    return (HEAD + ROW * ((n + 127) // 128 + 2) + 15) & ~15


def row_offset(r):
    return HEAD + ROW * r


def pack_row(tok, out, ordn):
    return struct.pack("<2I", (tok & 0xFFFFFF) | ((ordn & 0xFF) << 24), out & 0xFFFFFFFF)


def unpack_row(table, r):
    """-> (token position, output position, sequences before the row mod 256)"""
    w0, out = struct.unpack_from("<2I", table, row_offset(r))
    return w0 & 0xFFFFFF, out, w0 >> 24


def set_row(table, r, tok, out, ordn):
    """table: a bytearray"""
    table[row_offset(r):row_offset(r) + ROW] = pack_row(tok, out, ordn)


def pack_table(out_size, csize, nseq, rows):
    """rows: [(tok, out, sequences before)] without the end row; row 0 must be (0, 0, 0)"""
    t = struct.pack("<8I", MAGIC, out_size, csize, nseq, len(rows), 0, 0, 0)
    for tok, out, ordn in rows:
        t += pack_row(tok, out, ordn)
    t += pack_row(csize, out_size, nseq)
    return t.ljust((len(t) + 15) & ~15, b"\0")


def header(table):
    """-> (magic, out_size, csize, nseq, nrows)"""
    return struct.unpack_from("<5I", table, 0)


def is_valid(table):
    return struct.unpack_from("<I", table, 0)[0] == MAGIC


def table_bytes(nrows):
    """bytes a valid table of nrows rows uses"""
    return HEAD + ROW * (nrows + 1)
