/*
 * lz4frame.h -- LZ4 frame container API of the MI355X-native LZ4 codec (liblz4_amd).
 *
 * Drop-in declarations for the one-shot frame entry points of the reference library
 * (lz4/lz4 v1.10.0, lib/lz4frame.h); types have the reference's layout (they are ABI), each
 * prototype cites the reference declaration it replaces.  Frames written here are standard LZ4
 * frames (doc/lz4_Frame_format.md) and decode with any conforming decoder (`lz4 -d`); frames
 * written by the reference (`lz4 -B# -BI/-BD -BX`, LZ4F_compressFrame) decode here.
 *
 * The block payloads are compressed / decompressed by the gfx950 batch kernels (all blocks of a
 * frame in one launch), block checksums by the batched XXH32 kernel; the container fields, the
 * header checksum and the content checksum (one strictly serial XXH32 over the whole content,
 * lib/xxhash.c:352-389 - it cannot be split across lanes) are host C.
 *
 * Differences from the reference, all within the frame format:
 *  - Linked frames (the default, lz4frame.c:787-792) are compressed with all blocks in one launch
 *    too (the history of a block is source data); they are DECODED block after block, since a
 *    block needs the previous one's output (lz4frame.c:1901-1915) - ask for LZ4F_blockIndependent
 *    when decode speed matters.
 *  - compressionLevel >= 2 selects the HC kernel (lz4frame.c:943-958), lower values the fast one; negative
 *    "acceleration" levels are served by the fast kernel as level 0.
 *  - LZ4F_decompress is a streaming state machine like the reference's (lz4frame.c:1613-2060): input is taken item by
 *    item, never past the frame; the complete blocks it holds are decoded as one table on the device as soon as the
 *    caller's input runs dry, the end mark shows up or a batch is full (32 MiB of input, 1024 blocks or 128 MiB of
 *    output); a large batch is decoded beside the calls that hand out the batch before it.  Return values are the
 *    reference's size hints.  A stored block that arrives in pieces is handed on piece by piece as lz4frame.c:1790-1830
 *    does; a compressed block is delivered when it is complete.
 *  - The streaming compression context (LZ4F_compressBegin / Update / flush / End, lz4frame_stream_api.c) sends
 *    one block per call to the device; LZ4F_compressFrame sends the whole frame at once.
 *  - Dictionaries (LZ4F_CDict, *_usingDict, lz4frame.h:560-640): a frame begun with a dictionary compresses its first block
 *    (every block, when blocks are independent) with the dictionary's last 64 KB as history, one block per launch.
 */
#ifndef LZ4_AMD_LZ4FRAME_H
#define LZ4_AMD_LZ4FRAME_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef size_t LZ4F_errorCode_t;                                   /* lz4frame.h:105 */
unsigned    LZ4F_isError(LZ4F_errorCode_t code);                   /* lz4frame.h:107 */
const char* LZ4F_getErrorName(LZ4F_errorCode_t code);              /* lz4frame.h:108 */

/* lz4frame.h:123-198: parameter enums and structures (layout is ABI) */
typedef enum { LZ4F_default = 0, LZ4F_max64KB = 4, LZ4F_max256KB = 5, LZ4F_max1MB = 6, LZ4F_max4MB = 7 } LZ4F_blockSizeID_t;
typedef enum { LZ4F_blockLinked = 0, LZ4F_blockIndependent } LZ4F_blockMode_t;
typedef enum { LZ4F_noContentChecksum = 0, LZ4F_contentChecksumEnabled } LZ4F_contentChecksum_t;
typedef enum { LZ4F_noBlockChecksum = 0, LZ4F_blockChecksumEnabled } LZ4F_blockChecksum_t;
typedef enum { LZ4F_frame = 0, LZ4F_skippableFrame } LZ4F_frameType_t;

typedef struct {
    LZ4F_blockSizeID_t     blockSizeID;
    LZ4F_blockMode_t       blockMode;
    LZ4F_contentChecksum_t contentChecksumFlag;
    LZ4F_frameType_t       frameType;
    unsigned long long     contentSize;
    unsigned               dictID;
    LZ4F_blockChecksum_t   blockChecksumFlag;
} LZ4F_frameInfo_t;

typedef struct {
    LZ4F_frameInfo_t frameInfo;
    int      compressionLevel;
    unsigned autoFlush;
    unsigned favorDecSpeed;
    unsigned reserved[3];
} LZ4F_preferences_t;
#define LZ4F_INIT_FRAMEINFO   { LZ4F_max64KB, LZ4F_blockLinked, LZ4F_noContentChecksum, LZ4F_frame, 0ULL, 0U, LZ4F_noBlockChecksum }   /* lz4frame.h:185 */
#define LZ4F_INIT_PREFERENCES { LZ4F_INIT_FRAMEINFO, 0, 0u, 0u, { 0u, 0u, 0u } }                                                       /* lz4frame.h:200 */

#define LZ4F_VERSION 100                                            /* lz4frame.h:256 */
unsigned LZ4F_getVersion(void);                                     /* lz4frame.h:257 */

/* lz4frame.h:212, 224.  Result: frame size, or an error code (LZ4F_isError). */
size_t LZ4F_compressFrameBound(size_t srcSize, const LZ4F_preferences_t* preferencesPtr);
size_t LZ4F_compressFrame(void* dstBuffer, size_t dstCapacity, const void* srcBuffer, size_t srcSize,
                          const LZ4F_preferences_t* preferencesPtr);

/* streaming compression, lz4frame.h:240-366 */
typedef struct LZ4F_cctx_s LZ4F_cctx;
typedef LZ4F_cctx* LZ4F_compressionContext_t;
typedef struct { unsigned stableSrc; unsigned reserved[3]; } LZ4F_compressOptions_t;                 /* lz4frame.h:262-265 */
int              LZ4F_compressionLevel_max(void);                                                   /* lz4frame.h:240 */
LZ4F_errorCode_t LZ4F_createCompressionContext(LZ4F_cctx** cctxPtr, unsigned version);              /* lz4frame.h:274 */
LZ4F_errorCode_t LZ4F_freeCompressionContext(LZ4F_cctx* cctx);                                      /* lz4frame.h:275 */
size_t LZ4F_compressBegin(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_preferences_t* prefsPtr);   /* lz4frame.h:302 */
size_t LZ4F_compressBound(size_t srcSize, const LZ4F_preferences_t* prefsPtr);                      /* lz4frame.h:321 */
size_t LZ4F_compressUpdate(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const void* srcBuffer, size_t srcSize,
                           const LZ4F_compressOptions_t* cOptPtr);                                  /* lz4frame.h:335 */
size_t LZ4F_flush(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_compressOptions_t* cOptPtr);       /* lz4frame.h:349 */
size_t LZ4F_compressEnd(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_compressOptions_t* cOptPtr); /* lz4frame.h:363 */

/* lz4frame.h:366-382 */
typedef struct LZ4F_dctx_s LZ4F_dctx;
typedef LZ4F_dctx* LZ4F_decompressionContext_t;
typedef struct {
    unsigned stableDst;
    unsigned skipChecksums;
    unsigned reserved1;
    unsigned reserved0;
} LZ4F_decompressOptions_t;

LZ4F_errorCode_t LZ4F_createDecompressionContext(LZ4F_dctx** dctxPtr, unsigned version);   /* lz4frame.h:398 */
LZ4F_errorCode_t LZ4F_freeDecompressionContext(LZ4F_dctx* dctx);                           /* lz4frame.h:399 */
void             LZ4F_resetDecompressionContext(LZ4F_dctx* dctx);                          /* lz4frame.h:547 */

/* lz4frame.h:452: frame parameters from the header in src; *srcSizePtr receives the bytes consumed */
size_t LZ4F_getFrameInfo(LZ4F_dctx* dctx, LZ4F_frameInfo_t* frameInfoPtr, const void* srcBuffer, size_t* srcSizePtr);

/* lz4frame.h:497.  Returns 0 when the frame is completely decoded and delivered, a hint > 0
 * otherwise, or an error code.  *dstSizePtr / *srcSizePtr: in = capacity / available, out =
 * bytes written / consumed. */
size_t LZ4F_decompress(LZ4F_dctx* dctx, void* dstBuffer, size_t* dstSizePtr,
                       const void* srcBuffer, size_t* srcSizePtr, const LZ4F_decompressOptions_t* dOptPtr);

/* ---- the long tail (lz4frame.h:505-747; what tests/frametest.c links besides the calls above) */
size_t LZ4F_getBlockSize(LZ4F_blockSizeID_t blockSizeID);                                          /* lz4frame.h:702; lz4frame.c:333 */
size_t LZ4F_headerSize(const void* src, size_t srcSize);                                           /* lz4frame.h:429; lz4frame.c:1441 */
#define LZ4F_HEADER_SIZE_MIN  7
#define LZ4F_HEADER_SIZE_MAX 19
#define LZ4F_MIN_SIZE_TO_KNOW_HEADER_LENGTH 5
#define LZ4F_BLOCK_HEADER_SIZE 4
#define LZ4F_BLOCK_CHECKSUM_SIZE 4
#define LZ4F_CONTENT_CHECKSUM_SIZE 4
#define LZ4F_ENDMARK_SIZE 4
#define LZ4F_MAGICNUMBER 0x184D2204U
#define LZ4F_MAGIC_SKIPPABLE_START 0x184D2A50U
/* lz4frame.h:716 - bytes added to the frame as stored blocks (independent blocks only) */
size_t LZ4F_uncompressedUpdate(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const void* srcBuffer, size_t srcSize,
                               const LZ4F_compressOptions_t* cOptPtr);
/* dictionaries (lz4frame.h:560-640) */
typedef struct LZ4F_CDict_s LZ4F_CDict;
LZ4F_CDict* LZ4F_createCDict(const void* dictBuffer, size_t dictSize);
void        LZ4F_freeCDict(LZ4F_CDict* CDict);
size_t LZ4F_compressFrame_usingCDict(LZ4F_cctx* cctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                                     const LZ4F_CDict* cdict, const LZ4F_preferences_t* preferencesPtr);
size_t LZ4F_compressBegin_usingCDict(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_CDict* cdict, const LZ4F_preferences_t* prefsPtr);
size_t LZ4F_compressBegin_usingDict(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const void* dictBuffer, size_t dictSize, const LZ4F_preferences_t* prefsPtr);
/* reference lz4frame.c:824-836 (not in its header; lz4io.c declares it itself): the dictionary is the history of the frame's first block only */
size_t LZ4F_compressBegin_usingDictOnce(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const void* dictBuffer, size_t dictSize, const LZ4F_preferences_t* prefsPtr);
size_t LZ4F_decompress_usingDict(LZ4F_dctx* dctxPtr, void* dstBuffer, size_t* dstSizePtr, const void* srcBuffer, size_t* srcSizePtr,
                                 const void* dict, size_t dictSize, const LZ4F_decompressOptions_t* decompressOptionsPtr);
/* custom memory (lz4frame.h:712-747) */
typedef void* (*LZ4F_AllocFunction)(void* opaqueState, size_t size);
typedef void* (*LZ4F_CallocFunction)(void* opaqueState, size_t size);
typedef void  (*LZ4F_FreeFunction)(void* opaqueState, void* address);
typedef struct { LZ4F_AllocFunction customAlloc; LZ4F_CallocFunction customCalloc; LZ4F_FreeFunction customFree; void* opaqueState; } LZ4F_CustomMem;
LZ4F_cctx*  LZ4F_createCompressionContext_advanced(LZ4F_CustomMem customMem, unsigned version);
LZ4F_dctx*  LZ4F_createDecompressionContext_advanced(LZ4F_CustomMem customMem, unsigned version);
LZ4F_CDict* LZ4F_createCDict_advanced(LZ4F_CustomMem customMem, const void* dictBuffer, size_t dictSize);

/* lz4frame.h:656-686: error codes, in the reference's order */
typedef enum {
    LZ4F_OK_NoError = 0, LZ4F_ERROR_GENERIC, LZ4F_ERROR_maxBlockSize_invalid, LZ4F_ERROR_blockMode_invalid,
    LZ4F_ERROR_parameter_invalid, LZ4F_ERROR_compressionLevel_invalid, LZ4F_ERROR_headerVersion_wrong,
    LZ4F_ERROR_blockChecksum_invalid, LZ4F_ERROR_reservedFlag_set, LZ4F_ERROR_allocation_failed,
    LZ4F_ERROR_srcSize_tooLarge, LZ4F_ERROR_dstMaxSize_tooSmall, LZ4F_ERROR_frameHeader_incomplete,
    LZ4F_ERROR_frameType_unknown, LZ4F_ERROR_frameSize_wrong, LZ4F_ERROR_srcPtr_wrong,
    LZ4F_ERROR_decompressionFailed, LZ4F_ERROR_headerChecksum_invalid, LZ4F_ERROR_contentChecksum_invalid,
    LZ4F_ERROR_frameDecoding_alreadyStarted, LZ4F_ERROR_compressionState_uninitialized,
    LZ4F_ERROR_parameter_null, LZ4F_ERROR_io_write, LZ4F_ERROR_io_read, LZ4F_ERROR_maxCode
} LZ4F_errorCodes;
LZ4F_errorCodes LZ4F_getErrorCode(size_t functionResult);           /* lz4frame.h:689 */

#ifdef __cplusplus
}
#endif
#endif
