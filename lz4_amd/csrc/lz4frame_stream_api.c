/*
 * lz4frame_stream_api.c -- the reference's streaming frame COMPRESSION context (lib/lz4frame.h:262-366,
 * lz4frame.c:596-1265: LZ4F_createCompressionContext / compressBegin / compressBound / compressUpdate /
 * flush / compressEnd) on top of the GPU block codec.
 *
 * Same contract as the reference: input arrives in arbitrary pieces, is gathered into blocks of the frame's
 * block size, and every full block (or, with autoFlush / LZ4F_flush, every partial one) leaves as
 * [LE32 size | payload | optional XXH32]; blocks of a linked frame reference the 64 KB before them
 * (lz4frame.c:917-943, LZ4_compress_fast_continue there; the kernels' history mode here).  One block per
 * device round trip: the drop-in path.  The container fields and both checksums are host C.
 */
#include "../../include/lz4frame.h"
#include "../../include/lz4hc.h"
#include "lz4amd_internal.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ERR(e) ((size_t)-(ptrdiff_t)LZ4F_ERROR_##e)
#define WINDOW 65536u
#define BH 4u                /* block header / checksum / end mark size (lz4frame.c:271-273) */

#include "xxh32_host.h"

static size_t block_size_of(unsigned id)
{   /* lz4frame.c:333-341 */
    static const size_t sizes[4] = { 64u << 10, 256u << 10, 1u << 20, 4u << 20 };
    if (id == 0) id = LZ4F_max64KB;
    if (id < LZ4F_max64KB || id > LZ4F_max4MB) return 0;
    return sizes[id - LZ4F_max64KB];
}

/* ---- the context */
struct LZ4F_cctx_s {
    LZ4F_preferences_t prefs;
    unsigned version;
    int stage;                  /* 0: needs compressBegin, 1: inside a frame */
    size_t block_size;
    uint8_t* win;               /* [WINDOW bytes of history][block being gathered] */
    size_t hist, fill;          /* valid history bytes (end at win + WINDOW), bytes gathered */
    uint64_t total_in;
    xxh32_state xxh;
    LZ4F_CustomMem cmem;        /* lz4frame.h:712-727: the context and its buffer come from the caller's allocator when one is given */
    uint8_t* dict; size_t dict_len;     /* compressBegin_usingDict / usingCDict: the last 64 KB of the dictionary (a copy) */
    int dict_once;                      /* a raw dictionary buffer (usingDict / usingDictOnce) is the history of the frame's FIRST block only; a CDict that of every independent block (lz4frame.c:690-826) */
    int fill_raw;               /* the gathered bytes came through LZ4F_uncompressedUpdate: they leave as a stored block */
};
struct LZ4F_CDict_s { LZ4F_CustomMem cmem; uint8_t* content; size_t size; };

static void* cm_alloc(const LZ4F_CustomMem* m, size_t n, int zero)
{
    if (m->customCalloc && zero) return m->customCalloc(m->opaqueState, n);
    if (m->customAlloc) { void* p = m->customAlloc(m->opaqueState, n); if (p && zero) memset(p, 0, n); return p; }
    return zero ? calloc(1, n) : malloc(n);
}
static void cm_free(const LZ4F_CustomMem* m, void* p)
{
    if (!p) return;
    if (m->customFree) m->customFree(m->opaqueState, p); else free(p);
}

int LZ4F_compressionLevel_max(void) { return LZ4HC_CLEVEL_MAX; }

LZ4F_cctx* LZ4F_createCompressionContext_advanced(LZ4F_CustomMem customMem, unsigned version)
{   /* lz4frame.c:604-617 */
    LZ4F_cctx* const c = (LZ4F_cctx*)cm_alloc(&customMem, sizeof *c, 1);
    if (!c) return NULL;
    c->cmem = customMem;
    c->version = version;
    return c;
}
LZ4F_errorCode_t LZ4F_createCompressionContext(LZ4F_cctx** cctxPtr, unsigned version)
{   /* lz4frame.c:626-639 */
    LZ4F_CustomMem none;
    if (cctxPtr == NULL) return ERR(parameter_null);
    memset(&none, 0, sizeof none);
    *cctxPtr = LZ4F_createCompressionContext_advanced(none, version);
    return *cctxPtr ? 0 : ERR(allocation_failed);
}
LZ4F_errorCode_t LZ4F_freeCompressionContext(LZ4F_cctx* c)
{
    if (c) { const LZ4F_CustomMem m = c->cmem; cm_free(&m, c->win); cm_free(&m, c->dict); cm_free(&m, c); }
    return 0;
}

/* ---- dictionaries (lz4frame.c:531-594, 690-826): a CDict is the dictionary's last 64 KB; a frame that starts with one
 * compresses its first block (every block, when blocks are independent) with those bytes as history */
LZ4F_CDict* LZ4F_createCDict_advanced(LZ4F_CustomMem cmem, const void* dictBuffer, size_t dictSize)
{
    LZ4F_CDict* const cd = (LZ4F_CDict*)cm_alloc(&cmem, sizeof *cd, 1);
    if (!cd) return NULL;
    cd->cmem = cmem;
    if (dictSize > WINDOW) { dictBuffer = (const uint8_t*)dictBuffer + (dictSize - WINDOW); dictSize = WINDOW; }
    cd->content = (uint8_t*)cm_alloc(&cmem, dictSize ? dictSize : 1, 0);
    if (!cd->content) { cm_free(&cmem, cd); return NULL; }
    if (dictSize) memcpy(cd->content, dictBuffer, dictSize);
    cd->size = dictSize;
    return cd;
}
LZ4F_CDict* LZ4F_createCDict(const void* dictBuffer, size_t dictSize)
{
    LZ4F_CustomMem none;
    memset(&none, 0, sizeof none);
    return LZ4F_createCDict_advanced(none, dictBuffer, dictSize);
}
void LZ4F_freeCDict(LZ4F_CDict* cd)
{
    if (cd) { const LZ4F_CustomMem m = cd->cmem; cm_free(&m, cd->content); cm_free(&m, cd); }
}

static size_t bound_internal(size_t srcSize, const LZ4F_preferences_t* prefsPtr, size_t alreadyBuffered)
{   /* lz4frame.c:379-404 */
    LZ4F_preferences_t worst;
    memset(&worst, 0, sizeof worst);
    worst.frameInfo.contentChecksumFlag = LZ4F_contentChecksumEnabled;
    worst.frameInfo.blockChecksumFlag = LZ4F_blockChecksumEnabled;
    {   const LZ4F_preferences_t* const p = prefsPtr ? prefsPtr : &worst;
        const unsigned flush = p->autoFlush | (srcSize == 0);
        const size_t bs = block_size_of(p->frameInfo.blockSizeID);
        const size_t buffered = alreadyBuffered < bs - 1 ? alreadyBuffered : bs - 1;
        const size_t maxSrc = srcSize + buffered;
        const size_t nfull = maxSrc / bs, partial = maxSrc & (bs - 1), last = flush ? partial : 0;
        const size_t nblocks = nfull + (last > 0);
        return (BH + BH * (size_t)p->frameInfo.blockChecksumFlag) * nblocks + bs * nfull + last
             + BH + BH * (size_t)p->frameInfo.contentChecksumFlag;
    }
}
size_t LZ4F_compressBound(size_t srcSize, const LZ4F_preferences_t* prefsPtr)
{   /* lz4frame.c:419-424 */
    if (prefsPtr && prefsPtr->autoFlush) return bound_internal(srcSize, prefsPtr, 0);
    return bound_internal(srcSize, prefsPtr, (size_t)-1);
}

static size_t begin_internal(LZ4F_cctx* c, void* dstBuffer, size_t dstCapacity, const void* dict, size_t dictSize, const LZ4F_preferences_t* prefsPtr);
size_t LZ4F_compressBegin(LZ4F_cctx* c, void* dstBuffer, size_t dstCapacity, const LZ4F_preferences_t* prefsPtr)
{
    return begin_internal(c, dstBuffer, dstCapacity, NULL, 0, prefsPtr);
}
size_t LZ4F_compressBegin_usingDictOnce(LZ4F_cctx* c, void* dstBuffer, size_t dstCapacity, const void* dictBuffer, size_t dictSize, const LZ4F_preferences_t* prefsPtr)
{   /* lz4frame.c:824-836: the dictionary is the history of the frame's first block (the lz4 CLI's multi-threaded linked-block
     * compression starts every job with the 64 KB before it this way, lz4io.c:1118-1128) */
    const size_t r = begin_internal(c, dstBuffer, dstCapacity, dictBuffer, dictSize, prefsPtr);
    if (c && !LZ4F_isError(r)) c->dict_once = 1;
    return r;
}
size_t LZ4F_compressBegin_usingDict(LZ4F_cctx* c, void* dstBuffer, size_t dstCapacity, const void* dictBuffer, size_t dictSize, const LZ4F_preferences_t* prefsPtr)
{   /* lz4frame.c:838-849: the same thing ("this will only use the dictionary once") */
    return LZ4F_compressBegin_usingDictOnce(c, dstBuffer, dstCapacity, dictBuffer, dictSize, prefsPtr);
}
size_t LZ4F_compressBegin_usingCDict(LZ4F_cctx* c, void* dstBuffer, size_t dstCapacity, const LZ4F_CDict* cdict, const LZ4F_preferences_t* prefsPtr)
{   /* lz4frame.c:862-868 */
    return begin_internal(c, dstBuffer, dstCapacity, cdict ? cdict->content : NULL, cdict ? cdict->size : 0, prefsPtr);
}
static size_t begin_internal(LZ4F_cctx* c, void* dstBuffer, size_t dstCapacity, const void* dict, size_t dictSize, const LZ4F_preferences_t* prefsPtr)
{   /* lz4frame.c:690-826 */
    uint8_t* op = (uint8_t*)dstBuffer;
    if (c == NULL || dstBuffer == NULL) return ERR(parameter_null);
    if (dstCapacity < 19) return ERR(dstMaxSize_tooSmall);                 /* LZ4F_HEADER_SIZE_MAX */
    if (prefsPtr) c->prefs = *prefsPtr; else memset(&c->prefs, 0, sizeof c->prefs);
    if (c->prefs.frameInfo.blockSizeID == 0) c->prefs.frameInfo.blockSizeID = LZ4F_max64KB;
    c->block_size = block_size_of(c->prefs.frameInfo.blockSizeID);
    if (!c->block_size) return ERR(maxBlockSize_invalid);
    cm_free(&c->cmem, c->win);
    c->win = (uint8_t*)cm_alloc(&c->cmem, WINDOW + c->block_size, 0);
    if (!c->win) return ERR(allocation_failed);
    c->hist = c->fill = 0; c->total_in = 0; c->fill_raw = 0;
    cm_free(&c->cmem, c->dict); c->dict = NULL; c->dict_len = 0; c->dict_once = 0;
    if (dict && dictSize) {
        if (dictSize > WINDOW) { dict = (const uint8_t*)dict + (dictSize - WINDOW); dictSize = WINDOW; }
        c->dict = (uint8_t*)cm_alloc(&c->cmem, dictSize, 0);
        if (!c->dict) return ERR(allocation_failed);
        memcpy(c->dict, dict, dictSize); c->dict_len = dictSize;
        memcpy(c->win + WINDOW - dictSize, dict, dictSize); c->hist = dictSize;      /* the first block's history */
    }
    xxh32_reset(&c->xxh);
    /* header (lz4frame.c:779-813) */
    wr32(op, 0x184D2204u); op += 4;
    {   uint8_t* const desc = op;
        const LZ4F_frameInfo_t* f = &c->prefs.frameInfo;
        *op++ = (uint8_t)((1u << 6) | ((f->blockMode & 1u) << 5) | ((f->blockChecksumFlag & 1u) << 4)
                          | ((f->contentSize != 0) << 3) | ((f->contentChecksumFlag & 1u) << 2) | (f->dictID != 0));
        *op++ = (uint8_t)((unsigned)f->blockSizeID << 4);
        if (f->contentSize) { wr32(op, (uint32_t)f->contentSize); wr32(op + 4, (uint32_t)(f->contentSize >> 32)); op += 8; }
        if (f->dictID) { wr32(op, f->dictID); op += 4; }
        *op = (uint8_t)(xxh32_once(desc, (size_t)(op - desc)) >> 8); op++;
    }
    c->stage = 1;
    return (size_t)(op - (uint8_t*)dstBuffer);
}

/* one block out of the gathering buffer: [size | payload | checksum]; returns bytes written (0 on device failure) */
static size_t put_block(LZ4F_cctx* c, uint8_t* op)
{
    const size_t n = c->fill;
    uint8_t* const blk = c->win + WINDOW;
    const int linked = c->prefs.frameInfo.blockMode == LZ4F_blockLinked;
    uint8_t* const start = op;
    int cs = 0;
    /* lz4frame.c:943-958: levels >= LZ4HC_CLEVEL_MIN take the HC compressor; linked blocks see the 64 KB before them, the
     * blocks of an independent-block frame that was begun with a dictionary see the dictionary (lz4frame.c:917-943) */
    if (!linked && c->dict_len && !c->dict_once) { memcpy(c->win + WINDOW - c->dict_len, c->dict, c->dict_len); c->hist = c->dict_len; }
    if (!c->fill_raw)
        cs = lz4amd_compress_with_history(c->hist ? (const char*)blk - c->hist : NULL, (int)c->hist,
                                          (const char*)blk, (char*)op + BH, (int)n, (int)n - 1,
                                          c->prefs.compressionLevel >= LZ4HC_CLEVEL_MIN
                                              ? (c->prefs.compressionLevel | (c->prefs.favorDecSpeed ? LZ4AMD_HC_FAVOR_DEC_SPEED : 0))
                                              : (c->prefs.compressionLevel < 0 ? c->prefs.compressionLevel - 1 : 0));      /* lz4frame.c:713: favorDecSpeed; 924-927: a negative level is the acceleration -level + 1 */
    c->fill_raw = 0;
    if (cs <= 0 || (size_t)cs >= n) {                            /* lz4frame.c:896-899: stored raw */
        wr32(op, (uint32_t)n | 0x80000000u); memcpy(op + BH, blk, n); cs = (int)n;
    } else wr32(op, (uint32_t)cs);
    op += BH + (size_t)cs;
    if (c->prefs.frameInfo.blockChecksumFlag) { wr32(op, xxh32_once(start + BH, (size_t)cs)); op += BH; }   /* lz4frame.c:904 */
    /* slide: the last 64 KB of everything seen stay in front of the gathering area */
    if (linked) {
        const size_t total = c->hist + n, keep = total < WINDOW ? total : WINDOW;
        memmove(c->win + WINDOW - keep, blk + n - keep, keep);
        c->hist = keep;
    } else c->hist = 0;
    c->fill = 0;
    return (size_t)(op - start);
}

size_t LZ4F_compressUpdate(LZ4F_cctx* c, void* dstBuffer, size_t dstCapacity, const void* srcBuffer, size_t srcSize,
                           const LZ4F_compressOptions_t* cOptPtr)
{   /* lz4frame.c:989-1118 */
    const uint8_t* ip = (const uint8_t*)srcBuffer;
    uint8_t* op = (uint8_t*)dstBuffer;
    (void)cOptPtr;
    if (c == NULL || dstBuffer == NULL || (srcBuffer == NULL && srcSize)) return ERR(parameter_null);
    if (c->stage != 1) return ERR(compressionState_uninitialized);
    if (dstCapacity < bound_internal(srcSize, &c->prefs, c->fill)) return ERR(dstMaxSize_tooSmall);
    if (c->fill && c->fill_raw) op += put_block(c, op);          /* lz4frame.c:1013-1018: the kind of block changes: what was gathered leaves first */
    if (c->prefs.frameInfo.contentChecksumFlag) xxh32_update(&c->xxh, ip, srcSize);
    c->total_in += srcSize;
    while (srcSize) {
        const size_t room = c->block_size - c->fill, take = srcSize < room ? srcSize : room;
        memcpy(c->win + WINDOW + c->fill, ip, take);
        c->fill += take; ip += take; srcSize -= take;
        if (c->fill == c->block_size) op += put_block(c, op);
    }
    if (c->prefs.autoFlush && c->fill) op += put_block(c, op);
    return (size_t)(op - (uint8_t*)dstBuffer);
}

size_t LZ4F_uncompressedUpdate(LZ4F_cctx* c, void* dstBuffer, size_t dstCapacity, const void* srcBuffer, size_t srcSize,
                               const LZ4F_compressOptions_t* cOptPtr)
{   /* lz4frame.c:1139-1147: the bytes go into the frame as stored blocks (independent blocks only: a stored block leaves no
     * compression history behind) */
    const uint8_t* ip = (const uint8_t*)srcBuffer;
    uint8_t* op = (uint8_t*)dstBuffer;
    (void)cOptPtr;
    if (c == NULL || dstBuffer == NULL || (srcBuffer == NULL && srcSize)) return ERR(parameter_null);
    if (c->stage != 1) return ERR(compressionState_uninitialized);
    if (c->prefs.frameInfo.blockMode != LZ4F_blockIndependent) return ERR(blockMode_invalid);
    if (dstCapacity < bound_internal(srcSize, &c->prefs, c->fill)) return ERR(dstMaxSize_tooSmall);
    if (c->fill && !c->fill_raw) op += put_block(c, op);         /* what was gathered for compression leaves first */
    if (c->prefs.frameInfo.contentChecksumFlag) xxh32_update(&c->xxh, ip, srcSize);
    c->total_in += srcSize;
    while (srcSize) {
        const size_t room = c->block_size - c->fill, take = srcSize < room ? srcSize : room;
        memcpy(c->win + WINDOW + c->fill, ip, take);
        c->fill += take; ip += take; srcSize -= take; c->fill_raw = 1;
        if (c->fill == c->block_size) op += put_block(c, op);
    }
    if (c->prefs.autoFlush && c->fill) op += put_block(c, op);
    return (size_t)(op - (uint8_t*)dstBuffer);
}

size_t LZ4F_compressFrame_usingCDict(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const void* srcBuffer, size_t srcSize,
                                     const LZ4F_CDict* cdict, const LZ4F_preferences_t* preferencesPtr)
{   /* lz4frame.c:433-477: one frame through the streaming calls with autoFlush, the block size that fits the input and the
     * content size corrected; without a dictionary this is LZ4F_compressFrame (all blocks in one launch) */
    LZ4F_preferences_t prefs;
    uint8_t* const dst = (uint8_t*)dstBuffer;
    uint8_t* op = dst;
    size_t r;
    if (cdict == NULL || cdict->size == 0) return LZ4F_compressFrame(dstBuffer, dstCapacity, srcBuffer, srcSize, preferencesPtr);
    if (cctx == NULL) return ERR(parameter_null);
    if (preferencesPtr) prefs = *preferencesPtr; else memset(&prefs, 0, sizeof prefs);
    if (prefs.frameInfo.contentSize != 0) prefs.frameInfo.contentSize = (unsigned long long)srcSize;
    {   unsigned id = prefs.frameInfo.blockSizeID ? (unsigned)prefs.frameInfo.blockSizeID : LZ4F_max64KB, proposed = LZ4F_max64KB;
        while (id > proposed) { if (srcSize <= block_size_of(proposed)) { id = proposed; break; } proposed++; }    /* lz4frame.c:388-398 */
        prefs.frameInfo.blockSizeID = (LZ4F_blockSizeID_t)id; }
    prefs.autoFlush = 1;
    if (srcSize <= block_size_of(prefs.frameInfo.blockSizeID)) prefs.frameInfo.blockMode = LZ4F_blockIndependent;   /* lz4frame.c:441-442 */
    if (dstCapacity < LZ4F_compressFrameBound(srcSize, &prefs)) return ERR(dstMaxSize_tooSmall);
    r = LZ4F_compressBegin_usingCDict(cctx, op, dstCapacity, cdict, &prefs);
    if (LZ4F_isError(r)) return r;
    op += r;
    r = LZ4F_compressUpdate(cctx, op, dstCapacity - (size_t)(op - dst), srcBuffer, srcSize, NULL);
    if (LZ4F_isError(r)) return r;
    op += r;
    r = LZ4F_compressEnd(cctx, op, dstCapacity - (size_t)(op - dst), NULL);
    if (LZ4F_isError(r)) return r;
    op += r;
    return (size_t)(op - dst);
}

size_t LZ4F_flush(LZ4F_cctx* c, void* dstBuffer, size_t dstCapacity, const LZ4F_compressOptions_t* cOptPtr)
{   /* lz4frame.c:1160-1196 */
    (void)cOptPtr;
    if (c == NULL) return ERR(parameter_null);
    if (c->fill == 0) return 0;
    if (c->stage != 1) return ERR(compressionState_uninitialized);
    if (dstBuffer == NULL) return ERR(parameter_null);
    if (dstCapacity < c->fill + BH + BH) return ERR(dstMaxSize_tooSmall);
    return put_block(c, (uint8_t*)dstBuffer);
}

size_t LZ4F_compressEnd(LZ4F_cctx* c, void* dstBuffer, size_t dstCapacity, const LZ4F_compressOptions_t* cOptPtr)
{   /* lz4frame.c:1206-1247 */
    uint8_t* op = (uint8_t*)dstBuffer;
    size_t flushed;
    if (c == NULL || dstBuffer == NULL) return ERR(parameter_null);
    if (c->stage != 1) return ERR(compressionState_uninitialized);
    if (dstCapacity < bound_internal(0, &c->prefs, c->fill)) return ERR(dstMaxSize_tooSmall);
    flushed = LZ4F_flush(c, op, dstCapacity, cOptPtr);
    if (LZ4F_isError(flushed)) return flushed;
    op += flushed;
    wr32(op, 0); op += BH;
    if (c->prefs.frameInfo.contentChecksumFlag) { wr32(op, xxh32_digest(&c->xxh)); op += BH; }
    c->stage = 0;
    if (c->prefs.frameInfo.contentSize && c->prefs.frameInfo.contentSize != c->total_in) return ERR(frameSize_wrong);   /* lz4frame.c:1242-1245 */
    return (size_t)(op - (uint8_t*)dstBuffer);
}
