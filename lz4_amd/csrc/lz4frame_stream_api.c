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
};

int LZ4F_compressionLevel_max(void) { return LZ4HC_CLEVEL_MAX; }

LZ4F_errorCode_t LZ4F_createCompressionContext(LZ4F_cctx** cctxPtr, unsigned version)
{   /* lz4frame.c:626-639 */
    LZ4F_cctx* c;
    if (cctxPtr == NULL) return ERR(parameter_null);
    c = (LZ4F_cctx*)calloc(1, sizeof *c);
    if (!c) return ERR(allocation_failed);
    c->version = version;
    *cctxPtr = c;
    return 0;
}
LZ4F_errorCode_t LZ4F_freeCompressionContext(LZ4F_cctx* c)
{
    if (c) { free(c->win); free(c); }
    return 0;
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

size_t LZ4F_compressBegin(LZ4F_cctx* c, void* dstBuffer, size_t dstCapacity, const LZ4F_preferences_t* prefsPtr)
{   /* lz4frame.c:690-826 */
    uint8_t* op = (uint8_t*)dstBuffer;
    if (c == NULL || dstBuffer == NULL) return ERR(parameter_null);
    if (dstCapacity < 19) return ERR(dstMaxSize_tooSmall);                 /* LZ4F_HEADER_SIZE_MAX */
    if (prefsPtr) c->prefs = *prefsPtr; else memset(&c->prefs, 0, sizeof c->prefs);
    if (c->prefs.frameInfo.blockSizeID == 0) c->prefs.frameInfo.blockSizeID = LZ4F_max64KB;
    c->block_size = block_size_of(c->prefs.frameInfo.blockSizeID);
    if (!c->block_size) return ERR(maxBlockSize_invalid);
    {   uint8_t* const w = (uint8_t*)realloc(c->win, WINDOW + c->block_size);
        if (!w) return ERR(allocation_failed);
        c->win = w; }
    c->hist = c->fill = 0; c->total_in = 0;
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
    int cs;
    /* lz4frame.c:943-958: levels >= LZ4HC_CLEVEL_MIN take the HC compressor; linked blocks see the 64 KB before them */
    cs = lz4amd_compress_with_history(linked && c->hist ? (const char*)blk - c->hist : NULL, (int)c->hist,
                                      (const char*)blk, (char*)op + BH, (int)n, (int)n - 1,
                                      c->prefs.compressionLevel >= LZ4HC_CLEVEL_MIN ? c->prefs.compressionLevel : 0);
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
    }
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
