/*
 * lz4hc.h -- high-compression block API of the MI355X-native LZ4 codec (liblz4_amd).
 *
 * Drop-in declarations for the one-shot entry points of the reference's lib/lz4hc.h (v1.10.0);
 * each prototype cites the declaration it replaces.  Same names, argument meaning and return
 * conventions; host pointers in and out.  The search is the hash-chain match finder of the
 * reference's levels 3..9 re-designed for a GPU workgroup (lz4_amd/csrc/kernels/lz4_hc_kernel.h);
 * the bytes differ from the CPU library's, decode identically with any LZ4 decoder, and the size at
 * level 9 stays within the +-3 % window of the reference's (tests/test_gpu_hc.py).
 *
 * The streaming HC context (lz4hc.h:98-180) tracks where the previous data is and ships its last 64 KB with
 * every block as history (lz4hc_api.c).
 *
 * Levels 10-12 (lz4hc.c:92-106, LZ4HC_compress_optimal 1823-2130) run an optimal parse on the device: the level-9 search
 * result of every position, then the cheapest sequence boundaries by price (kernels/lz4_hc_kernel.h: hc_parse_strip_opt).
 * Levels 10 / 11 / 12 search 96 / 512 / 2048 candidates per position (the reference: 96 / 512 / 16384) and share level 10's
 * 64-byte sufficient length; lz4amd_last_notice() says so after a call with level 11 or 12.  Levels 1-2 are the reference's LZ4MID
 * (lz4hc.c:93-95, 472-773): two tables keyed by 4 and by 7 bytes, one candidate each (kernels/lz4_hc_kernel.h: hc_search_mid).
 */
#ifndef LZ4_AMD_LZ4HC_H
#define LZ4_AMD_LZ4HC_H

#include "lz4.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference lz4hc.h:47-50 */
#define LZ4HC_CLEVEL_MIN         2
#define LZ4HC_CLEVEL_DEFAULT     9
#define LZ4HC_CLEVEL_OPT_MIN    10
#define LZ4HC_CLEVEL_MAX        12

/* reference lz4hc.h:252: size a caller must provide for an external HC state */
#define LZ4_STREAMHC_MINSIZE  262200

/* lz4hc.h:66.  Returns the number of bytes written to dst, 0 if the block does not fit in
 * dstCapacity.  Always succeeds when dstCapacity >= LZ4_compressBound(srcSize). */
int LZ4_compress_HC(const char* src, char* dst, int srcSize, int dstCapacity, int compressionLevel);

int LZ4_sizeofStateHC(void);                                                        /* lz4hc.h:79 */
/* lz4hc.h:80.  The caller-provided state is not needed by the device path (head table and chains
 * live in LDS / device scratch); it is accepted for ABI compatibility and must be non-NULL and
 * 8-byte aligned as in the reference (lz4hc.c:1506). */
int LZ4_compress_HC_extStateHC(void* stateHC, const char* src, char* dst, int srcSize, int maxDstSize, int compressionLevel);

/* ---- streaming (reference lz4hc.h:98-180, 275, 356-389) */
typedef union LZ4_streamHC_u {
    char minStateSize[LZ4_STREAMHC_MINSIZE];                /* lz4hc.h:252-256: the size is ABI */
    struct { const char* dictionary; unsigned dictSize; int compressionLevel;
             signed char favorDecSpeed, dirty; /* (named as in lz4hc.h:246-248; a context here is never dirty: nothing survives a failed call) */ } internal_donotuse;
} LZ4_streamHC_t;
LZ4_streamHC_t* LZ4_createStreamHC(void);                                          /* lz4hc.h:109 */
int             LZ4_freeStreamHC(LZ4_streamHC_t* streamHCPtr);                     /* lz4hc.h:110 */
LZ4_streamHC_t* LZ4_initStreamHC(void* buffer, size_t size);                       /* lz4hc.h:275 */
void            LZ4_resetStreamHC_fast(LZ4_streamHC_t* streamHCPtr, int compressionLevel);   /* lz4hc.h:157 */
void            LZ4_resetStreamHC(LZ4_streamHC_t* streamHCPtr, int compressionLevel);        /* lz4hc.h:322 */
void            LZ4_setCompressionLevel(LZ4_streamHC_t* streamHCPtr, int compressionLevel);  /* lz4hc.h:356 */
int             LZ4_loadDictHC(LZ4_streamHC_t* streamHCPtr, const char* dictionary, int dictSize);   /* lz4hc.h:158 */
int             LZ4_compress_HC_continue(LZ4_streamHC_t* streamHCPtr, const char* src, char* dst,
                                         int srcSize, int maxDstSize);             /* lz4hc.h:160 */
int             LZ4_saveDictHC(LZ4_streamHC_t* streamHCPtr, char* safeBuffer, int maxDictSize);      /* lz4hc.h:178 */

/* ---- the long tail (lz4_amd/csrc/lz4_compat_api.c) */
int  LZ4_compress_HC_extStateHC_fastReset(void* state, const char* src, char* dst, int srcSize, int dstCapacity, int compressionLevel);   /* lz4hc.h:397 */
void LZ4_attach_HC_dictionary(LZ4_streamHC_t* working_stream, const LZ4_streamHC_t* dictionary_stream);      /* lz4hc.h:403 */
void LZ4_favorDecompressionSpeed(LZ4_streamHC_t* LZ4_streamHCPtr, int favor);    /* lz4hc.h:364: acts on levels 10-12 (no offsets < 8, lengths 19..36 cut to 18) */
/* lz4hc.h:89, 170: as much of src as fits targetDstSize (prefixes compressed on the device, bisection) */
int  LZ4_compress_HC_destSize(void* stateHC, const char* src, char* dst, int* srcSizePtr, int targetDstSize, int compressionLevel);
int  LZ4_compress_HC_continue_destSize(LZ4_streamHC_t* LZ4_streamHCPtr, const char* src, char* dst, int* srcSizePtr, int targetDstSize);
/* deprecated names, thin wrappers (lz4hc.h:290-352) */
int  LZ4_compressHC(const char* source, char* dest, int inputSize);
int  LZ4_compressHC_limitedOutput(const char* source, char* dest, int inputSize, int maxOutputSize);
int  LZ4_compressHC2(const char* source, char* dest, int inputSize, int compressionLevel);
int  LZ4_compressHC2_limitedOutput(const char* source, char* dest, int inputSize, int maxOutputSize, int compressionLevel);
int  LZ4_compressHC_withStateHC(void* state, const char* source, char* dest, int inputSize);
int  LZ4_compressHC_limitedOutput_withStateHC(void* state, const char* source, char* dest, int inputSize, int maxOutputSize);
int  LZ4_compressHC2_withStateHC(void* state, const char* source, char* dest, int inputSize, int compressionLevel);
int  LZ4_compressHC2_limitedOutput_withStateHC(void* state, const char* source, char* dest, int inputSize, int maxOutputSize, int compressionLevel);
int  LZ4_compressHC_continue(LZ4_streamHC_t* LZ4_streamHCPtr, const char* source, char* dest, int inputSize);
int  LZ4_compressHC_limitedOutput_continue(LZ4_streamHC_t* LZ4_streamHCPtr, const char* source, char* dest, int inputSize, int maxOutputSize);
void* LZ4_createHC(const char* inputBuffer);
int   LZ4_freeHC(void* LZ4HC_Data);
char* LZ4_slideInputBufferHC(void* LZ4HC_Data);
int   LZ4_compressHC2_continue(void* LZ4HC_Data, const char* source, char* dest, int inputSize, int compressionLevel);
int   LZ4_compressHC2_limitedOutput_continue(void* LZ4HC_Data, const char* source, char* dest, int inputSize, int maxOutputSize, int compressionLevel);
int   LZ4_sizeofStreamStateHC(void);
int   LZ4_resetStreamStateHC(void* state, char* inputBuffer);

#ifdef __cplusplus
}
#endif
#endif
