/*
 * lz4_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the LZ4 block codec, XXH32 and the LZ4 frame
 * container, used as the checker for the HIP path.  Nothing in the shipped
 * library (lz4_amd/) may include, link or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity pin: every function here is checked against the real reference
 * (compiled from /root/reference into oracle/_ref/ by oracle/Makefile) and
 * against the committed golden vectors in tests/golden/ (see
 * tests/test_oracle_vs_reference.py, tests/golden/make_golden.py).
 */
#ifndef LZ4_ORACLE_H
#define LZ4_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LZ4O_MAX_INPUT_SIZE 0x7E000000

/* lz4.h:215  LZ4_COMPRESSBOUND */
int lz4o_compress_bound(int n);

/* lz4.c:1382 LZ4_compress_fast_extState + lz4.c:930 LZ4_compress_generic_validated
 * (noDict; byU16 below 64KB+11, byU32/hash5 above).  Byte-identical output to the
 * reference on little-endian 64-bit hosts. Returns bytes written, 0 on failure. */
int lz4o_compress_fast(const uint8_t* src, uint8_t* dst, int n, int cap, int accel);
int lz4o_compress_default(const uint8_t* src, uint8_t* dst, int n, int cap);

/* lz4.c:2023 LZ4_decompress_generic (decode_full_block), restated with the rules of the
 * safe loop (lz4.c:2215-2435) applied to every sequence.  `prefix` = number of valid
 * history bytes immediately before dst (0 = LZ4_decompress_safe, lz4.c:2451;
 * >0 = LZ4_decompress_safe_withPrefix64k / _withSmallPrefix, lz4.c:2479,2504).
 * Returns decoded size, or a negative value on malformed input / dst too small. */
int lz4o_decompress_safe(const uint8_t* src, uint8_t* dst, int csize, int cap);
int lz4o_decompress_safe_prefix(const uint8_t* src, uint8_t* dst, int csize, int cap,
                                size_t prefix);

/* xxhash.c:392 XXH32 */
uint32_t lz4o_xxh32(const void* data, size_t len, uint32_t seed);

/* Frame container (doc/lz4_Frame_format.md; lz4frame.c:690-813 header, 883 makeBlock,
 * 1206 compressEnd).  blockSizeID 4..7; flags as in LZ4F_preferences_t.
 * Independent-block mode only on the compress side (linked mode is validated through
 * the decoder).  Returns frame size or 0 on error. */
size_t lz4o_frame_bound(size_t n, int blockSizeID, int blockChecksum, int contentChecksum);
size_t lz4o_frame_compress(uint8_t* dst, size_t cap, const uint8_t* src, size_t n,
                           int blockSizeID, int blockChecksum, int contentChecksum,
                           int contentSizeFlag);
/* Decodes one whole frame (linked or independent blocks, optional checksums).
 * Returns decoded size, or (size_t)-1 on any format/checksum error.
 * *consumed (optional) receives the number of source bytes of the frame. */
size_t lz4o_frame_decompress(uint8_t* dst, size_t cap, const uint8_t* src, size_t n,
                             size_t* consumed);

#ifdef __cplusplus
}
#endif
#endif
