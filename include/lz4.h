/*
 * lz4.h -- block API of the MI355X-native LZ4 codec (liblz4_amd).
 *
 * Drop-in declarations for the block-codec entry points of the reference library
 * (lz4/lz4 v1.10.0, lib/lz4.h); each prototype cites the reference declaration it replaces.
 * Same names, argument meaning and return conventions; host pointers in and out.  The work is
 * done by hand-written gfx950 kernels (see include/lz4amd.h for the batch interface the
 * kernels are really built for).  Compressed bytes may differ from the CPU library's but are
 * legal LZ4 blocks and decode to the same data with any conforming decoder.
 *
 * Streaming (lz4.h:314-560): LZ4_stream_t / LZ4_streamDecode_t contexts that track up to 64 KB of
 * history between calls, each block still a round trip through HBM (lz4_stream_api.c).
 *
 * The long tail of the reference header is here too (csrc/lz4_compat_api.c): LZ4_compress_destSize (device compressions of
 * prefixes), LZ4_attach_dictionary, and - host code by their contract, they do not know where their input or output
 * ends - LZ4_decompress_safe_partial and the deprecated LZ4_decompress_fast family.
 */
#ifndef LZ4_AMD_LZ4_H
#define LZ4_AMD_LZ4_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference lz4.h:131-140 */
#define LZ4_VERSION_MAJOR    1
#define LZ4_VERSION_MINOR   10
#define LZ4_VERSION_RELEASE  0
#define LZ4_VERSION_NUMBER (LZ4_VERSION_MAJOR *100*100 + LZ4_VERSION_MINOR *100 + LZ4_VERSION_RELEASE)
#define LZ4_VERSION_STRING "1.10.0"

/* reference lz4.h:214-215 */
#define LZ4_MAX_INPUT_SIZE        0x7E000000
#define LZ4_COMPRESSBOUND(isize)  ((unsigned)(isize) > (unsigned)LZ4_MAX_INPUT_SIZE ? 0 : (isize) + ((isize)/255) + 16)

/* reference lz4.h:673-675 */
#define LZ4_DISTANCE_MAX 65535

/* reference lz4.h:729: size a caller must provide for an external compression state */
#define LZ4_STREAM_MINSIZE  ((1UL << 14) + 32)

int         LZ4_versionNumber(void);                                              /* lz4.h:142 */
const char* LZ4_versionString(void);                                              /* lz4.h:143 */

/* lz4.h:191.  Returns the number of bytes written to dst (<= dstCapacity), 0 if the block does
 * not fit.  Always succeeds when dstCapacity >= LZ4_compressBound(srcSize). */
int LZ4_compress_default(const char* src, char* dst, int srcSize, int dstCapacity);

/* lz4.h:208.  Returns the number of bytes decoded (<= dstCapacity), or a negative value if src
 * is malformed or dst too small.  Never reads outside src[0,compressedSize) nor writes outside
 * dst[0,dstCapacity). */
int LZ4_decompress_safe(const char* src, char* dst, int compressedSize, int dstCapacity);
/* Two deliberate differences from the reference's safe decoder, both on the strict side of the block format document
 * (neither occurs in anything a LZ4 compressor emits; tests/test_gpu_parity.py pins them):
 *  - a sequence with offset 0 is malformed (the reference only checks `match < lowPrefix`, lz4.c:2356, and copies
 *    undefined bytes);
 *  - the end-of-block rules are applied to every sequence, not only behind the reference's fast loop
 *    (lz4.c:2279, 2312-2318, 2423): a block the reference's fast loop lets through by accident is rejected here.
 * Against the real reference on 800 mutated blocks the verdicts differ on fewer than 2 % (measured: 0). */

/* lz4.h:549.  As LZ4_decompress_safe, with up to 64 KB of history in [dictStart, dictStart+dictSize)
 * (what lz4frame passes for linked blocks, lz4frame.c:1901). */
int LZ4_decompress_safe_usingDict(const char* src, char* dst, int compressedSize, int dstCapacity,
                                  const char* dictStart, int dictSize);

int LZ4_compressBound(int inputSize);                                              /* lz4.h:226 */
int LZ4_compress_fast(const char* src, char* dst, int srcSize, int dstCapacity, int acceleration);   /* lz4.h:236; acceleration 1: every second position of a block of 64 KB or more is probed, 2 and above: every fourth */
int LZ4_sizeofState(void);                                                         /* lz4.h:245 */
int LZ4_compress_fast_extState(void* state, const char* src, char* dst, int srcSize, int dstCapacity, int acceleration);   /* lz4.h:246 */

/* ---- streaming compression (reference lz4.h:314-450).  The context remembers WHERE the previous data
 * is (the caller keeps it in place, as with the reference) and hands its last 64 KB to the device as
 * history of the next block. */
typedef union LZ4_stream_u {
    char minStateSize[LZ4_STREAM_MINSIZE];                  /* lz4.h:729-733: the size is ABI */
    struct { const char* dictionary; unsigned dictSize; } internal_donotuse;
} LZ4_stream_t;
LZ4_stream_t* LZ4_createStream(void);                                              /* lz4.h:331 */
int           LZ4_freeStream(LZ4_stream_t* streamPtr);                             /* lz4.h:332 */
LZ4_stream_t* LZ4_initStream(void* stateBuffer, size_t size);                      /* lz4.h:750 */
void          LZ4_resetStream_fast(LZ4_stream_t* streamPtr);                       /* lz4.h:358 */
void          LZ4_resetStream(LZ4_stream_t* streamPtr);                            /* lz4.h:876 */
int           LZ4_loadDict(LZ4_stream_t* streamPtr, const char* dictionary, int dictSize);   /* lz4.h:371 */
int           LZ4_compress_fast_continue(LZ4_stream_t* streamPtr, const char* src, char* dst,
                                         int srcSize, int dstCapacity, int acceleration);    /* lz4.h:441 */
int           LZ4_saveDict(LZ4_stream_t* streamPtr, char* safeBuffer, int maxDictSize);      /* lz4.h:450 */

/* ---- streaming decompression (reference lz4.h:457-535, state as lz4.h:757-770) */
#define LZ4_STREAMDECODE_MINSIZE 32
typedef union LZ4_streamDecode_u {
    char minStateSize[LZ4_STREAMDECODE_MINSIZE];
    struct { const unsigned char* externalDict; const unsigned char* prefixEnd; size_t extDictSize; size_t prefixSize; } internal_donotuse;
} LZ4_streamDecode_t;
LZ4_streamDecode_t* LZ4_createStreamDecode(void);                                  /* lz4.h:465 */
int                 LZ4_freeStreamDecode(LZ4_streamDecode_t* LZ4_stream);          /* lz4.h:466 */
int                 LZ4_setStreamDecode(LZ4_streamDecode_t* LZ4_streamDecode, const char* dictionary, int dictSize);   /* lz4.h:477 */
int                 LZ4_decoderRingBufferSize(int maxBlockSize);                   /* lz4.h:490 */
int                 LZ4_decompress_safe_continue(LZ4_streamDecode_t* LZ4_streamDecode, const char* src, char* dst,
                                                 int srcSize, int dstCapacity);   /* lz4.h:531 */

/* ---- the long tail (lz4_amd/csrc/lz4_compat_api.c): what the reference's own tests and CLI link against besides the
 * hot path.  Device-backed unless marked "host": */
int  LZ4_compress_fast_extState_fastReset(void* state, const char* src, char* dst, int srcSize, int dstCapacity, int acceleration);   /* lz4.h:611 */
void LZ4_attach_dictionary(LZ4_stream_t* workingStream, const LZ4_stream_t* dictionaryStream);      /* lz4.h:640; lz4.c:1658 */
int  LZ4_loadDictSlow(LZ4_stream_t* streamPtr, const char* dictionary, int dictSize);               /* lz4.h:380 */
/* lz4.h:568, lz4.c:1506: as much of src as fits targetDstSize; *srcSizePtr = bytes consumed.  Found by compressing
 * prefixes on the device (bisection), so a call costs ~log2(srcSize) launches when the input does not fit whole. */
int  LZ4_compress_destSize(const char* src, char* dst, int* srcSizePtr, int targetDstSize);
int  LZ4_compress_destSize_extState(void* state, const char* src, char* dst, int* srcSizePtr, int targetDstSize, int acceleration);   /* lz4.h:582 */
int  LZ4_decompress_safe_withPrefix64k(const char* src, char* dst, int compressedSize, int maxDstSize);       /* lz4.c:2479 */
/* host (no device kernel: the caller does not say how large the block's output is), written from the block format document: */
int  LZ4_decompress_safe_partial(const char* src, char* dst, int srcSize, int targetOutputSize, int dstCapacity);   /* lz4.h:291; lz4.c:2459 */
int  LZ4_decompress_safe_partial_usingDict(const char* src, char* dst, int compressedSize, int targetOutputSize, int maxOutputSize,
                                           const char* dictStart, int dictSize);                                   /* lz4.h:553 */
/* deprecated and host (lz4.h:806-826: the input's size is unknown, the caller vouches for it): */
int  LZ4_decompress_fast(const char* src, char* dst, int originalSize);
int  LZ4_decompress_fast_continue(LZ4_streamDecode_t* LZ4_streamDecode, const char* src, char* dst, int originalSize);
int  LZ4_decompress_fast_usingDict(const char* src, char* dst, int originalSize, const char* dictStart, int dictSize);
int  LZ4_decompress_fast_withPrefix64k(const char* src, char* dst, int originalSize);
/* deprecated names, thin wrappers (lz4.h:784-845) */
int  LZ4_compress(const char* src, char* dest, int srcSize);
int  LZ4_compress_limitedOutput(const char* src, char* dest, int srcSize, int maxOutputSize);
int  LZ4_compress_withState(void* state, const char* source, char* dest, int inputSize);
int  LZ4_compress_limitedOutput_withState(void* state, const char* source, char* dest, int inputSize, int maxOutputSize);
int  LZ4_compress_continue(LZ4_stream_t* LZ4_streamPtr, const char* source, char* dest, int inputSize);
int  LZ4_compress_limitedOutput_continue(LZ4_stream_t* LZ4_streamPtr, const char* source, char* dest, int inputSize, int maxOutputSize);
int  LZ4_uncompress(const char* source, char* dest, int outputSize);
int  LZ4_uncompress_unknownOutputSize(const char* source, char* dest, int isize, int maxOutputSize);
void* LZ4_create(char* inputBuffer);
int   LZ4_sizeofStreamState(void);
int   LZ4_resetStreamState(void* state, char* inputBuffer);
char* LZ4_slideInputBuffer(void* state);
/* in-place (de)compression (lz4.h:637-678): both work here as in the reference - a block is staged through the device,
 * so the source is read completely before the first output byte is written */
#define LZ4_DISTANCE_MAX 65535
#define LZ4_DECOMPRESS_INPLACE_MARGIN(compressedSize)          (((compressedSize) >> 8) + 32)
#define LZ4_DECOMPRESS_INPLACE_BUFFER_SIZE(decompressedSize)   ((decompressedSize) + LZ4_DECOMPRESS_INPLACE_MARGIN(decompressedSize))
#define LZ4_COMPRESS_INPLACE_MARGIN                           (LZ4_DISTANCE_MAX + 32)
#define LZ4_COMPRESS_INPLACE_BUFFER_SIZE(maxCompressedSize)   ((maxCompressedSize) + LZ4_COMPRESS_INPLACE_MARGIN)

#ifdef __cplusplus
}
#endif
#endif
