/*
 * lz4hc_api.c -- the reference's one-shot high-compression C ABI (lib/lz4hc.h) on top of the GPU
 * batch codec.  Same names, argument meaning and return conventions as the reference
 * (lz4hc.c:1519 LZ4_compress_HC, 1503 LZ4_compress_HC_extStateHC, 1486 LZ4_sizeofStateHC): host
 * pointers in and out, the block makes a round trip through HBM.  No CPU codec: without a usable
 * HIP device the calls return 0 after a message on stderr.
 */
#include "../../include/lz4hc.h"
#include "../../include/lz4amd.h"
#include "lz4amd_internal.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int LZ4_compress_HC(const char* src, char* dst, int srcSize, int dstCapacity, int compressionLevel)
{
    lz4amd_set_notice(compressionLevel > 10 ? "LZ4_compress_HC: levels 11-12 search 512 / 2048 candidates per position (the reference: 512 / 16384) and use the 64-byte sufficient length of level 10" : "");
    if (srcSize < 0 || (unsigned)srcSize > (unsigned)LZ4_MAX_INPUT_SIZE) return 0;   /* lz4hc.c:1403 */
    if (dst == NULL || dstCapacity <= 0) return 0;
    if (src == NULL && srcSize != 0) return 0;
    return lz4amd_run_one(LZ4AMD_OP_COMPRESS_HC, src, dst, srcSize, dstCapacity, compressionLevel, 0);
}

int LZ4_sizeofStateHC(void) { return LZ4_STREAMHC_MINSIZE; }

int LZ4_compress_HC_extStateHC(void* stateHC, const char* src, char* dst, int srcSize, int maxDstSize, int compressionLevel)
{
    if (stateHC == NULL || ((uintptr_t)stateHC & (sizeof(void*) - 1)) != 0) return 0;   /* lz4hc.c:1506-1508 */
    return LZ4_compress_HC(src, dst, srcSize, maxDstSize, compressionLevel);
}

/* ------------------------------------------------------------------ streaming (lz4hc.c:1540-1760)
 * As with LZ4_stream_t (lz4_stream_api.c) the context is only the LOCATION of the previous data: the
 * head table and chains are rebuilt on the device per block, from the history shipped with it. */
#define WINDOW 65536u

LZ4_streamHC_t* LZ4_initStreamHC(void* buffer, size_t size)
{   /* lz4hc.c:1560-1573 */
    LZ4_streamHC_t* const s = (LZ4_streamHC_t*)buffer;
    if (buffer == NULL || size < sizeof(LZ4_streamHC_t) || ((size_t)buffer & (sizeof(void*) - 1))) return NULL;
    memset(s, 0, sizeof *s);
    s->internal_donotuse.compressionLevel = LZ4HC_CLEVEL_DEFAULT;
    return s;
}
LZ4_streamHC_t* LZ4_createStreamHC(void)
{
    LZ4_streamHC_t* const s = (LZ4_streamHC_t*)malloc(sizeof(LZ4_streamHC_t));
    if (s) LZ4_initStreamHC(s, sizeof *s);
    return s;
}
int LZ4_freeStreamHC(LZ4_streamHC_t* s) { free(s); return 0; }
void LZ4_setCompressionLevel(LZ4_streamHC_t* s, int level)
{   /* lz4hc.c:1601-1606 */
    if (level < 1) level = LZ4HC_CLEVEL_DEFAULT;
    if (level > LZ4HC_CLEVEL_MAX) level = LZ4HC_CLEVEL_MAX;
    s->internal_donotuse.compressionLevel = level;
}
void LZ4_resetStreamHC(LZ4_streamHC_t* s, int level) { if (LZ4_initStreamHC(s, sizeof *s)) LZ4_setCompressionLevel(s, level); }
void LZ4_resetStreamHC_fast(LZ4_streamHC_t* s, int level) { LZ4_resetStreamHC(s, level); }

int LZ4_loadDictHC(LZ4_streamHC_t* s, const char* dictionary, int dictSize)
{   /* lz4hc.c:1615-1640: the last 64 KB count */
    int const level = s->internal_donotuse.compressionLevel;
    LZ4_resetStreamHC(s, level);
    if (dictionary == NULL || dictSize <= 0) return 0;
    if ((unsigned)dictSize > WINDOW) { dictionary += (unsigned)dictSize - WINDOW; dictSize = (int)WINDOW; }
    s->internal_donotuse.dictionary = dictionary;
    s->internal_donotuse.dictSize = (unsigned)dictSize;
    return dictSize;
}

int LZ4_compress_HC_continue(LZ4_streamHC_t* s, const char* src, char* dst, int srcSize, int maxDstSize)
{   /* lz4hc.c:1666-1716 */
    const char* hist;
    unsigned hsz;
    int r;
    if (s == NULL || srcSize < 0) return 0;
    hist = s->internal_donotuse.dictionary; hsz = s->internal_donotuse.dictSize;
    if (hist && src < hist + hsz && src + srcSize > hist) {          /* overlap invalidates (part of) the dictionary, lz4hc.c:1693-1703 */
        const char* const srcEnd = src + srcSize;
        if (srcEnd >= hist + hsz) { hist = NULL; hsz = 0; }
        else { hsz = (unsigned)((hist + hsz) - srcEnd); hist = srcEnd; if (hsz < 4) { hist = NULL; hsz = 0; } }
    }
    r = lz4amd_compress_with_history(hist, (int)hsz, src, dst, srcSize, maxDstSize,
                                     s->internal_donotuse.compressionLevel | (s->internal_donotuse.favorDecSpeed ? LZ4AMD_HC_FAVOR_DEC_SPEED : 0));
    if (hist && hist + hsz == src) {                                 /* contiguous: the window slides over both */
        unsigned long long total = (unsigned long long)hsz + (unsigned)srcSize;
        if (total > WINDOW) { hist += total - WINDOW; total = WINDOW; }
        s->internal_donotuse.dictionary = hist; s->internal_donotuse.dictSize = (unsigned)total;
    } else {
        unsigned keep = (unsigned)srcSize;
        s->internal_donotuse.dictionary = src;
        if (keep > WINDOW) { s->internal_donotuse.dictionary = src + (keep - WINDOW); keep = WINDOW; }
        s->internal_donotuse.dictSize = keep;
    }
    return r;
}

int LZ4_saveDictHC(LZ4_streamHC_t* s, char* safeBuffer, int maxDictSize)
{   /* lz4hc.c:1736-1760 */
    unsigned n;
    if (s == NULL || maxDictSize < 0) return 0;
    n = s->internal_donotuse.dictSize;
    if (n > (unsigned)maxDictSize) n = (unsigned)maxDictSize;
    if (n > WINDOW) n = WINDOW;
    if (n < 4 || safeBuffer == NULL) n = 0;                          /* lz4hc.c:1743 */
    if (n) memmove(safeBuffer, s->internal_donotuse.dictionary + s->internal_donotuse.dictSize - n, n);
    s->internal_donotuse.dictionary = safeBuffer;
    s->internal_donotuse.dictSize = n;
    return (int)n;
}
