/*
 * lz4_stream_api.c -- the reference's streaming contexts (lib/lz4.h:314-560) on top of the GPU block codec.
 *
 * A streaming context in the reference is a hash table plus the location of the previous data
 * (lz4.c:1543-1800 for compression, 2585-2668 for decompression).  Here the tables live on the device
 * and are rebuilt per call, so a context is only the location: the last 64 KB before the new block are
 * shipped to the device as the block's history (the kernels' prefix mode).  Same calling rules as the
 * reference: previously processed data must stay where it was (or be moved with LZ4_saveDict /
 * announced with LZ4_setStreamDecode).  One block per call = one round trip through HBM: this is the
 * drop-in path, not the fast one (include/lz4amd.h is).
 */
#include "../../include/lz4.h"
#include "lz4amd_internal.h"
#include <stdlib.h>
#include <string.h>

#define WINDOW 65536

/* ------------------------------------------------------------------ compression */
LZ4_stream_t* LZ4_initStream(void* buffer, size_t size)
{   /* lz4.c:1560-1568 */
    if (buffer == NULL || size < sizeof(LZ4_stream_t) || ((size_t)buffer & (sizeof(void*) - 1))) return NULL;
    memset(buffer, 0, sizeof(LZ4_stream_t));
    return (LZ4_stream_t*)buffer;
}
LZ4_stream_t* LZ4_createStream(void)
{
    LZ4_stream_t* const s = (LZ4_stream_t*)malloc(sizeof(LZ4_stream_t));
    if (s) LZ4_initStream(s, sizeof *s);
    return s;
}
int  LZ4_freeStream(LZ4_stream_t* s) { free(s); return 0; }
void LZ4_resetStream(LZ4_stream_t* s) { if (s) memset(s, 0, sizeof *s); }
void LZ4_resetStream_fast(LZ4_stream_t* s) { LZ4_resetStream(s); }

int LZ4_loadDict(LZ4_stream_t* s, const char* dictionary, int dictSize)
{   /* lz4.c:1626-1675: only the last 64 KB count; returns the size kept */
    if (s == NULL) return 0;
    LZ4_resetStream(s);
    if (dictionary == NULL || dictSize < 4) return 0;            /* lz4.c:1650 HASH_UNIT */
    if (dictSize > WINDOW) { dictionary += dictSize - WINDOW; dictSize = WINDOW; }
    s->internal_donotuse.dictionary = dictionary;
    s->internal_donotuse.dictSize = (unsigned)dictSize;
    return dictSize;
}

int LZ4_compress_fast_continue(LZ4_stream_t* s, const char* src, char* dst, int srcSize, int dstCapacity, int acceleration)
{   /* lz4.c:1707-1781 */
    const char* hist;
    unsigned hsz;
    int r;
    if (s == NULL) return 0;
    hist = s->internal_donotuse.dictionary; hsz = s->internal_donotuse.dictSize;
    /* a source that overlaps the dictionary invalidates the overlapped part (lz4.c:1737-1747) */
    if (hist && src < hist + hsz && src + (srcSize > 0 ? srcSize : 0) > hist) {
        const char* const srcEnd = src + srcSize;
        if (srcEnd >= hist + hsz) { hist = NULL; hsz = 0; }
        else { hsz = (unsigned)((hist + hsz) - srcEnd); hist = srcEnd; if (hsz < 4) { hist = NULL; hsz = 0; } }
    }
    r = lz4amd_compress_with_history(hist, (int)hsz, src, dst, srcSize, dstCapacity, acceleration > 1 ? -(acceleration > 65537 ? 65537 : acceleration) : 0);
    /* what the next call may reference: the block, plus what precedes it if it is contiguous (prefix
     * mode, lz4.c:1750-1757); otherwise only the block (lz4.c:1776-1779) */
    if (hist && hist + hsz == src) {
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

int LZ4_saveDict(LZ4_stream_t* s, char* safeBuffer, int maxDictSize)
{   /* lz4.c:1806-1833 */
    unsigned n;
    if (s == NULL || maxDictSize < 0) return 0;
    n = s->internal_donotuse.dictSize;
    if (n > (unsigned)maxDictSize) n = (unsigned)maxDictSize;
    if (n > WINDOW) n = WINDOW;
    if (safeBuffer == NULL) n = 0;
    if (n) memmove(safeBuffer, s->internal_donotuse.dictionary + s->internal_donotuse.dictSize - n, n);
    s->internal_donotuse.dictionary = safeBuffer;
    s->internal_donotuse.dictSize = n;
    return (int)n;
}

/* ------------------------------------------------------------------ decompression */
LZ4_streamDecode_t* LZ4_createStreamDecode(void) { return (LZ4_streamDecode_t*)calloc(1, sizeof(LZ4_streamDecode_t)); }
int LZ4_freeStreamDecode(LZ4_streamDecode_t* sd) { free(sd); return 0; }

int LZ4_setStreamDecode(LZ4_streamDecode_t* sd, const char* dictionary, int dictSize)
{   /* lz4.c:2598-2609 */
    if (sd == NULL) return 0;
    sd->internal_donotuse.prefixSize = dictionary && dictSize > 0 ? (size_t)dictSize : 0;
    sd->internal_donotuse.prefixEnd = sd->internal_donotuse.prefixSize ? (const unsigned char*)dictionary + dictSize : (const unsigned char*)dictionary;
    sd->internal_donotuse.externalDict = NULL;
    sd->internal_donotuse.extDictSize = 0;
    return 1;
}

int LZ4_decoderRingBufferSize(int maxBlockSize)
{   /* lz4.c:2622-2628, LZ4_DECODER_RING_BUFFER_SIZE lz4.h:491 */
    if (maxBlockSize < 0 || maxBlockSize > LZ4_MAX_INPUT_SIZE) return 0;
    if (maxBlockSize < 16) maxBlockSize = 16;
    return 65536 + 14 + maxBlockSize;
}

int LZ4_decompress_safe_continue(LZ4_streamDecode_t* sd, const char* src, char* dst, int srcSize, int dstCapacity)
{   /* lz4.c:2631-2668: first call / rolling prefix / wrapped or switched buffer */
    int r;
    if (sd == NULL) return -1;
    if (sd->internal_donotuse.prefixSize == 0) {
        r = LZ4_decompress_safe(src, dst, srcSize, dstCapacity);
        if (r <= 0) return r;
        sd->internal_donotuse.prefixSize = (size_t)r;
        sd->internal_donotuse.prefixEnd = (const unsigned char*)dst + r;
    } else if (sd->internal_donotuse.prefixEnd == (const unsigned char*)dst) {
        const size_t ps = sd->internal_donotuse.prefixSize, es = sd->internal_donotuse.extDictSize;
        if (ps >= WINDOW - 1 || es == 0) {
            const size_t use = ps > WINDOW ? WINDOW : ps;
            r = LZ4_decompress_safe_usingDict(src, dst, srcSize, dstCapacity, (const char*)dst - use, (int)use);
        } else {
            /* prefix shorter than the window plus an older segment elsewhere (the reference's doubleDict,
             * lz4.c:2545-2556): the last 64 KB of history are gathered into one buffer for the device */
            size_t from_ext = WINDOW - ps;
            char* h;
            if (from_ext > es) from_ext = es;
            h = (char*)malloc(from_ext + ps);
            if (!h) return -1;
            memcpy(h, sd->internal_donotuse.externalDict + es - from_ext, from_ext);
            memcpy(h + from_ext, (const char*)dst - ps, ps);
            r = LZ4_decompress_safe_usingDict(src, dst, srcSize, dstCapacity, h, (int)(from_ext + ps));
            free(h);
        }
        if (r <= 0) return r;
        sd->internal_donotuse.prefixSize += (size_t)r;
        sd->internal_donotuse.prefixEnd += r;
    } else {
        sd->internal_donotuse.extDictSize = sd->internal_donotuse.prefixSize;
        sd->internal_donotuse.externalDict = sd->internal_donotuse.prefixEnd - sd->internal_donotuse.extDictSize;
        {   const size_t es = sd->internal_donotuse.extDictSize, use = es > WINDOW ? WINDOW : es;
            r = LZ4_decompress_safe_usingDict(src, dst, srcSize, dstCapacity,
                                              (const char*)sd->internal_donotuse.externalDict + es - use, (int)use); }
        if (r <= 0) return r;
        sd->internal_donotuse.prefixSize = (size_t)r;
        sd->internal_donotuse.prefixEnd = (const unsigned char*)dst + r;
    }
    return r;
}
