/*
 * lz4frame_api.c -- the LZ4 frame container (doc/lz4_Frame_format.md) around the GPU batch block
 * codec: one-shot LZ4F_compressFrame (lib/lz4frame.c:484 -> 428 -> compressBegin 690 / makeBlock
 * 883 / compressEnd 1206) and LZ4F_decompress (lz4frame.c:1613-2116; header 1346-1437).
 *
 * Host C only handles the container: magic, FLG/BD, optional content size / dictID, header
 * checksum, per-block size fields, end mark, content checksum.  All blocks of a frame go to the
 * device in ONE block table (lz4amd_batch.c): compression, decompression and block checksums are
 * single launches over the whole frame.  The content checksum is one serial XXH32 over the whole
 * content (xxhash.c:352-389; the recurrence cannot be split), computed on the calling thread while
 * the GPU works.  There is no CPU codec here: without a HIP device every call returns an error.
 */
#include "../../include/lz4frame.h"
#include "../../include/lz4amd.h"
#include "lz4amd_internal.h"
#include "lz4amd_ffi.h"
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ERR(code) ((size_t)-(ptrdiff_t)(LZ4F_ERROR_##code))
#define MAGIC 0x184D2204u              /* lz4frame.c:242 */
#define MAGIC_SKIP 0x184D2A50u         /* lz4frame.c:243, low 4 bits free */
#define MAX_HEADER 19                  /* lz4frame.c:246 */

extern pthread_mutex_t lz4amd_default_lock;        /* lz4_api.c */

/* ---------------------------------------------------------------- errors */
static const char* const k_error_names[] = {
    "OK_NoError", "ERROR_GENERIC", "ERROR_maxBlockSize_invalid", "ERROR_blockMode_invalid",
    "ERROR_parameter_invalid", "ERROR_compressionLevel_invalid", "ERROR_headerVersion_wrong",
    "ERROR_blockChecksum_invalid", "ERROR_reservedFlag_set", "ERROR_allocation_failed",
    "ERROR_srcSize_tooLarge", "ERROR_dstMaxSize_tooSmall", "ERROR_frameHeader_incomplete",
    "ERROR_frameType_unknown", "ERROR_frameSize_wrong", "ERROR_srcPtr_wrong",
    "ERROR_decompressionFailed", "ERROR_headerChecksum_invalid", "ERROR_contentChecksum_invalid",
    "ERROR_frameDecoding_alreadyStarted", "ERROR_compressionState_uninitialized",
    "ERROR_parameter_null", "ERROR_io_write", "ERROR_io_read", "ERROR_maxCode" };

unsigned LZ4F_isError(LZ4F_errorCode_t code) { return code > (size_t)-(ptrdiff_t)LZ4F_ERROR_maxCode; }   /* lz4frame.c:293-296 */
LZ4F_errorCodes LZ4F_getErrorCode(size_t r) { return LZ4F_isError(r) ? (LZ4F_errorCodes)(-(ptrdiff_t)r) : LZ4F_OK_NoError; }
const char* LZ4F_getErrorName(LZ4F_errorCode_t code)
{ return LZ4F_isError(code) ? k_error_names[-(ptrdiff_t)code] : "Unspecified error code"; }     /* lz4frame.c:298-303 */
unsigned LZ4F_getVersion(void) { return LZ4F_VERSION; }

/* ---------------------------------------------------------------- XXH32 (host: header / content checksum) */
#define P1 0x9E3779B1u
#define P2 0x85EBCA77u
#define P3 0xC2B2AE3Du
#define P4 0x27D4EB2Fu
#define P5 0x165667B1u
static uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static void wr32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t xxh32(const uint8_t* p, size_t len)
{   /* xxhash.c:352-389 stripes, 291-348 tail and avalanche; seed 0 */
    const uint8_t* const end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0u - P1;
        do {
            v1 = rotl(v1 + rd32(p) * P2, 13) * P1;      v2 = rotl(v2 + rd32(p + 4) * P2, 13) * P1;
            v3 = rotl(v3 + rd32(p + 8) * P2, 13) * P1;  v4 = rotl(v4 + rd32(p + 12) * P2, 13) * P1;
            p += 16;
        } while (p + 16 <= end);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    } else h = P5;
    h += (uint32_t)len;
    while (p + 4 <= end) { h = rotl(h + rd32(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl(h + (*p++) * P5, 11) * P1; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

static size_t block_size_of(unsigned id)
{   /* lz4frame.c:333-341 */
    static const size_t sizes[4] = { 64u << 10, 256u << 10, 1u << 20, 4u << 20 };
    if (id == 0) id = LZ4F_max64KB;
    if (id < LZ4F_max64KB || id > LZ4F_max4MB) return 0;
    return sizes[id - LZ4F_max64KB];
}
static unsigned optimal_bsid(unsigned requested, size_t srcSize)
{   /* lz4frame.c:388-398: the smallest block size that holds the input in one block */
    unsigned proposed = LZ4F_max64KB;
    while (requested > proposed) { if (srcSize <= block_size_of(proposed)) return proposed; proposed++; }
    return requested;
}

/* ---------------------------------------------------------------- device helpers */
typedef struct { void* in; size_t in_cap; void* out; size_t out_cap; } dev_stage;
static dev_stage g_stage;                           /* guarded by lz4amd_default_lock */
static int stage_fit(void** buf, size_t* cap, size_t need)
{
    if (need <= *cap) return 0;
    lz4amd_hip_free(*buf); *cap = 0;
    *buf = lz4amd_hip_malloc(need + (need >> 3) + 4096);
    if (!*buf) return -1;
    *cap = need + (need >> 3) + 4096;
    return 0;
}

/* ---------------------------------------------------------------- compression */
size_t LZ4F_compressFrameBound(size_t srcSize, const LZ4F_preferences_t* prefs)
{   /* lz4frame.c:406-416 with autoFlush: every block may be stored raw */
    LZ4F_preferences_t p;
    size_t bs, nb;
    if (prefs) p = *prefs; else memset(&p, 0, sizeof p);
    bs = block_size_of(optimal_bsid(p.frameInfo.blockSizeID ? p.frameInfo.blockSizeID : LZ4F_max64KB, srcSize));
    if (!bs) return ERR(maxBlockSize_invalid);
    nb = (srcSize + bs - 1) / bs;
    return MAX_HEADER + srcSize + nb * (4 + (p.frameInfo.blockChecksumFlag ? 4 : 0)) + 4 + (p.frameInfo.contentChecksumFlag ? 4 : 0);
}

size_t LZ4F_compressFrame(void* dstBuffer, size_t dstCapacity, const void* srcBuffer, size_t srcSize,
                          const LZ4F_preferences_t* prefs)
{
    LZ4F_preferences_t p;
    uint8_t* const dst = (uint8_t*)dstBuffer;
    const uint8_t* const src = (const uint8_t*)srcBuffer;
    uint8_t* op = dst;
    size_t bs, nb, stride, i, result = ERR(GENERIC);
    unsigned bsid;
    lz4amd_ctx* ctx;
    lz4amd_plan *cplan = NULL, *xplan = NULL;
    const void** d_src = NULL; void** d_dst = NULL; int *sizes = NULL, *caps = NULL, *csz = NULL, *sums = NULL, *pres = NULL;
    uint32_t content_sum = 0;
    int linked;

    if (prefs) p = *prefs; else memset(&p, 0, sizeof p);
    if (!dst || (!src && srcSize)) return ERR(parameter_null);
    bsid = optimal_bsid(p.frameInfo.blockSizeID ? p.frameInfo.blockSizeID : LZ4F_max64KB, srcSize);
    bs = block_size_of(bsid);
    if (!bs) return ERR(maxBlockSize_invalid);
    if (p.frameInfo.contentSize != 0) p.frameInfo.contentSize = srcSize;       /* lz4frame.c:445-446: auto-correct */
    if (dstCapacity < LZ4F_compressFrameBound(srcSize, &p)) return ERR(dstMaxSize_tooSmall);
    nb = (srcSize + bs - 1) / bs;
    linked = p.frameInfo.blockMode == LZ4F_blockLinked && nb > 1;          /* lz4frame.c:441-442: one block is independent */

    /* -- header (lz4frame.c:779-813) */
    wr32(op, MAGIC); op += 4;
    {   uint8_t* const desc = op;
        *op++ = (uint8_t)((1u << 6) | ((linked ? 0u : 1u) << 5) | ((p.frameInfo.blockChecksumFlag & 1u) << 4)
                          | ((p.frameInfo.contentSize != 0) << 3) | ((p.frameInfo.contentChecksumFlag & 1u) << 2)
                          | (p.frameInfo.dictID != 0));
        *op++ = (uint8_t)(bsid << 4);
        if (p.frameInfo.contentSize) { wr32(op, (uint32_t)p.frameInfo.contentSize); wr32(op + 4, (uint32_t)(p.frameInfo.contentSize >> 32)); op += 8; }
        if (p.frameInfo.dictID) { wr32(op, p.frameInfo.dictID); op += 4; }
        *op = (uint8_t)(xxh32(desc, (size_t)(op - desc)) >> 8); op++;
    }
    if (nb == 0) goto finish_frame;

    /* -- all blocks in one block table on the device */
    stride = (bs + bs / 255 + 16 + 255) & ~(size_t)255;
    d_src = (const void**)malloc(nb * sizeof *d_src); d_dst = (void**)malloc(nb * sizeof *d_dst);
    sizes = (int*)malloc(nb * sizeof *sizes); caps = (int*)malloc(nb * sizeof *caps);
    csz = (int*)malloc(nb * sizeof *csz); sums = (int*)malloc(nb * sizeof *sums); pres = (int*)malloc(nb * sizeof *pres);
    if (!d_src || !d_dst || !sizes || !caps || !csz || !sums || !pres) { result = ERR(allocation_failed); goto done_unlocked; }

    pthread_mutex_lock(&lz4amd_default_lock);
    ctx = lz4amd_default_ctx();
    if (!ctx) goto done;
    if (stage_fit(&g_stage.in, &g_stage.in_cap, srcSize + 64) || stage_fit(&g_stage.out, &g_stage.out_cap, nb * stride)) { result = ERR(allocation_failed); goto done; }
    if (lz4amd_hip_h2d(g_stage.in, src, srcSize, NULL)) goto done;
    for (i = 0; i < nb; i++) {
        const size_t chunk = (i + 1 < nb) ? bs : srcSize - i * bs;
        d_src[i] = (const char*)g_stage.in + i * bs; sizes[i] = (int)chunk;
        d_dst[i] = (char*)g_stage.out + i * stride; caps[i] = (int)chunk - 1;      /* lz4frame.c:891-899: must gain a byte */
        if (caps[i] < 1) caps[i] = 1;
        pres[i] = linked ? (int)(i * bs < 65536 ? i * bs : 65536) : 0;     /* the history is the source itself */
    }
    if (p.compressionLevel >= 2) {
        /* lz4frame.c:943-958 LZ4F_selectCompression: levels >= LZ4HC_CLEVEL_MIN take the HC compressor */
        if (lz4amd_plan_create_compress_hc_prefix(ctx, &cplan, (int)nb, d_src, sizes, d_dst, caps, linked ? pres : NULL, p.compressionLevel)) goto done;
    } else
    if (lz4amd_plan_create_compress_prefix(ctx, &cplan, (int)nb, d_src, sizes, d_dst, caps, linked ? pres : NULL)) goto done;
    if (lz4amd_plan_launch(cplan, NULL)) goto done;
    if (p.frameInfo.contentChecksumFlag) content_sum = xxh32(src, srcSize);      /* the host hashes while the GPU compresses */
    if (lz4amd_plan_results(cplan, csz, NULL)) goto done;
    for (i = 0; i < nb; i++) if (csz[i] <= 0 || csz[i] >= sizes[i]) csz[i] = 0;   /* stored raw */
    if (p.frameInfo.blockChecksumFlag) {          /* lz4frame.c:904: XXH32 of the block as stored */
        for (i = 0; i < nb; i++) { if (csz[i]) { d_src[i] = d_dst[i]; caps[i] = csz[i]; } else caps[i] = sizes[i]; }
        if (lz4amd_plan_create(ctx, &xplan, LZ4AMD_OP_XXH32, (int)nb, d_src, caps, NULL, NULL, 0)) goto done;
        if (lz4amd_plan_launch(xplan, NULL) || lz4amd_plan_results(xplan, sums, NULL)) goto done;
    }
    for (i = 0; i < nb; i++) {
        const uint32_t n = csz[i] ? (uint32_t)csz[i] : (uint32_t)sizes[i];
        wr32(op, csz[i] ? n : (n | 0x80000000u)); op += 4;
        if (csz[i]) { if (lz4amd_hip_d2h(op, (char*)g_stage.out + i * stride, n, NULL)) goto done; }
        else memcpy(op, src + i * bs, n);
        op += n;
        if (p.frameInfo.blockChecksumFlag) { wr32(op, (uint32_t)sums[i]); op += 4; }
    }
    if (lz4amd_hip_sync(NULL)) goto done;
    pthread_mutex_unlock(&lz4amd_default_lock);
    lz4amd_plan_destroy(cplan); lz4amd_plan_destroy(xplan); cplan = xplan = NULL;
    goto finish_frame_free;
done:
    pthread_mutex_unlock(&lz4amd_default_lock);
done_unlocked:
    lz4amd_plan_destroy(cplan); lz4amd_plan_destroy(xplan);
    free(d_src); free(d_dst); free(sizes); free(caps); free(csz); free(sums); free(pres);
    return result;
finish_frame_free:
    free(d_src); free(d_dst); free(sizes); free(caps); free(csz); free(sums); free(pres);
finish_frame:
    if (nb == 0 && p.frameInfo.contentChecksumFlag) content_sum = xxh32(src, 0);
    wr32(op, 0); op += 4;                                                     /* end mark, lz4frame.c:1222 */
    if (p.frameInfo.contentChecksumFlag) { wr32(op, content_sum); op += 4; }   /* lz4frame.c:1225-1231 */
    return (size_t)(op - dst);
}

/* ---------------------------------------------------------------- decompression */
struct LZ4F_dctx_s {
    uint8_t* in; size_t in_size, in_cap;          /* the frame's bytes so far */
    uint8_t* out; size_t out_size, out_pos;       /* decoded content, and how much of it was delivered */
    int header_done, frame_done, decoded;
    size_t header_size, frame_size;               /* frame_size: total bytes of the frame once known */
    LZ4F_frameInfo_t info;
    size_t block_max;
    size_t scan_pos;                              /* next block header to look at */
};

LZ4F_errorCode_t LZ4F_createDecompressionContext(LZ4F_dctx** dctxPtr, unsigned version)
{   /* lz4frame.c:1284-1310 */
    if (!dctxPtr) return ERR(parameter_null);
    (void)version;
    *dctxPtr = (LZ4F_dctx*)calloc(1, sizeof **dctxPtr);
    return *dctxPtr ? 0 : ERR(allocation_failed);
}
void LZ4F_resetDecompressionContext(LZ4F_dctx* d)
{
    if (!d) return;
    free(d->out); d->out = NULL; d->out_size = d->out_pos = 0;
    d->in_size = 0; d->header_done = d->frame_done = d->decoded = 0;
    d->header_size = d->frame_size = d->scan_pos = 0;
}
LZ4F_errorCode_t LZ4F_freeDecompressionContext(LZ4F_dctx* d)
{
    if (d) { free(d->in); free(d->out); free(d); }
    return 0;
}

/* parse the header at p (n bytes available).  Returns header size, 0 if more bytes are needed, or an error. */
static size_t parse_header(const uint8_t* p, size_t n, LZ4F_frameInfo_t* info, size_t* block_max)
{   /* lz4frame.c:1346-1437 */
    uint32_t magic;
    unsigned flg, bd, version, bsid;
    size_t hs;
    if (n < 4) return 0;
    magic = rd32(p);
    memset(info, 0, sizeof *info);
    if ((magic & 0xFFFFFFF0u) == MAGIC_SKIP) {
        if (n < 8) return 0;
        info->frameType = LZ4F_skippableFrame;
        info->contentSize = rd32(p + 4);           /* size of the user data that follows */
        return 8;
    }
    if (magic != MAGIC) return ERR(frameType_unknown);
    if (n < 7) return 0;
    flg = p[4]; bd = p[5];
    version = (flg >> 6) & 3;
    if (version != 1) return ERR(headerVersion_wrong);
    if (flg & 2) return ERR(reservedFlag_set);
    if ((bd & 0x80) || (bd & 0x0F)) return ERR(reservedFlag_set);
    bsid = (bd >> 4) & 7;
    if (bsid < 4) return ERR(maxBlockSize_invalid);
    hs = 7 + ((flg & 8) ? 8 : 0) + ((flg & 1) ? 4 : 0);
    if (n < hs) return 0;
    if (p[hs - 1] != (uint8_t)(xxh32(p + 4, hs - 5) >> 8)) return ERR(headerChecksum_invalid);
    info->blockSizeID = (LZ4F_blockSizeID_t)bsid;
    info->blockMode = (flg & 0x20) ? LZ4F_blockIndependent : LZ4F_blockLinked;
    info->blockChecksumFlag = (flg & 0x10) ? LZ4F_blockChecksumEnabled : LZ4F_noBlockChecksum;
    info->contentChecksumFlag = (flg & 4) ? LZ4F_contentChecksumEnabled : LZ4F_noContentChecksum;
    info->frameType = LZ4F_frame;
    if (flg & 8) info->contentSize = (unsigned long long)rd32(p + 6) | ((unsigned long long)rd32(p + 10) << 32);
    if (flg & 1) info->dictID = rd32(p + hs - 5);
    *block_max = block_size_of(bsid);
    return hs;
}

size_t LZ4F_getFrameInfo(LZ4F_dctx* d, LZ4F_frameInfo_t* info, const void* srcBuffer, size_t* srcSizePtr)
{   /* lz4frame.c:1464-1512 (one-shot form: the header must be in srcBuffer) */
    size_t bm = 0, hs;
    if (!d || !info || !srcSizePtr) return ERR(parameter_null);
    if (d->header_done) { *info = d->info; *srcSizePtr = 0; return 1; }
    hs = parse_header((const uint8_t*)srcBuffer, *srcSizePtr, info, &bm);
    if (LZ4F_isError(hs)) { *srcSizePtr = 0; return hs; }
    if (hs == 0) { *srcSizePtr = 0; return ERR(frameHeader_incomplete); }
    *srcSizePtr = 0;                       /* nothing consumed: LZ4F_decompress will read the header again */
    return 4;
}

/* decode the complete frame held in d->in into d->out */
static size_t decode_buffered_frame(LZ4F_dctx* d, int skip_checksums)
{
    const uint8_t* const base = d->in;
    size_t pos = d->header_size, nb = 0, i, out_total = 0, result = ERR(GENERIC);
    const int bchk = d->info.blockChecksumFlag == LZ4F_blockChecksumEnabled;
    const int linked = d->info.blockMode == LZ4F_blockLinked;
    lz4amd_ctx* ctx;
    lz4amd_plan *dplan = NULL, *xplan = NULL;
    const void** d_src = NULL; void** d_dst = NULL; int *sizes = NULL, *caps = NULL, *res = NULL, *sums = NULL, *prefix = NULL;
    size_t* in_off = NULL; uint8_t* raw = NULL;
    size_t ncomp = 0, in_bytes = 0;

    /* pass 1: count blocks */
    for (pos = d->header_size;;) { const uint32_t f = rd32(base + pos); if (!f) break; pos += 4 + (f & 0x7FFFFFFFu) + (bchk ? 4 : 0); nb++; }
    if (nb == 0) { d->out = (uint8_t*)malloc(1); d->out_size = 0; goto checksum; }
    d_src = (const void**)malloc(nb * sizeof *d_src); d_dst = (void**)malloc(nb * sizeof *d_dst);
    sizes = (int*)malloc(nb * sizeof *sizes); caps = (int*)malloc(nb * sizeof *caps); res = (int*)malloc(nb * sizeof *res);
    sums = (int*)malloc(nb * sizeof *sums); prefix = (int*)malloc(nb * sizeof *prefix);
    in_off = (size_t*)malloc(nb * sizeof *in_off); raw = (uint8_t*)malloc(nb);
    if (!d_src || !d_dst || !sizes || !caps || !res || !sums || !prefix || !in_off || !raw) { result = ERR(allocation_failed); goto done_unlocked; }
    for (pos = d->header_size, i = 0; i < nb; i++) {
        const uint32_t f = rd32(base + pos);
        sizes[i] = (int)(f & 0x7FFFFFFFu); raw[i] = (uint8_t)(f >> 31); in_off[i] = pos + 4;
        pos += 4 + (size_t)sizes[i] + (bchk ? 4 : 0);
    }
    in_bytes = pos;

    pthread_mutex_lock(&lz4amd_default_lock);
    ctx = lz4amd_default_ctx();
    if (!ctx) goto done;
    if (stage_fit(&g_stage.in, &g_stage.in_cap, in_bytes + 64) || stage_fit(&g_stage.out, &g_stage.out_cap, nb * d->block_max + 64)) { result = ERR(allocation_failed); goto done; }
    if (lz4amd_hip_h2d(g_stage.in, base, in_bytes, NULL)) goto done;
    /* block checksums: XXH32 of every block as stored (lz4frame.c:1878), one launch */
    if (bchk && !skip_checksums) {
        for (i = 0; i < nb; i++) d_src[i] = (const char*)g_stage.in + in_off[i];
        if (lz4amd_plan_create(ctx, &xplan, LZ4AMD_OP_XXH32, (int)nb, d_src, sizes, NULL, NULL, 0)) goto done;
        if (lz4amd_plan_launch(xplan, NULL) || lz4amd_plan_results(xplan, sums, NULL)) goto done;
        for (i = 0; i < nb; i++) if ((uint32_t)sums[i] != rd32(base + in_off[i] + (size_t)sizes[i])) { result = ERR(blockChecksum_invalid); goto done; }
    }
    /* compressed blocks -> one block table; every block owns a block_max slot of the output.
     * Linked frames (lz4frame.c:1901-1915: the previous 64 KB of output are the dictionary) need
     * the blocks decoded in order with a packed output: the table then runs on ONE workgroup. */
    if (linked) {
        /* packed output needs the decoded sizes in advance only for raw blocks; compressed ones
         * are decoded one after the other, each told how much history precedes it */
        size_t o = 0;
        for (i = 0; i < nb; i++) {
            d_dst[i] = (char*)g_stage.out + o;
            if (raw[i]) {
                if ((size_t)sizes[i] > d->block_max) { result = ERR(decompressionFailed); goto done; }
                if (lz4amd_hip_h2d(d_dst[i], base + in_off[i], (size_t)sizes[i], NULL)) goto done;
                res[i] = sizes[i];
            } else {
                const void* s1 = (const char*)g_stage.in + in_off[i];
                int cap1 = (int)d->block_max, pre1 = (int)(o < 65536 ? o : 65536);
                lz4amd_plan* p1 = NULL;
                if (lz4amd_plan_create_prefix(ctx, &p1, 1, &s1, &sizes[i], &d_dst[i], &cap1, &pre1)) goto done;
                if (lz4amd_plan_launch(p1, NULL) || lz4amd_plan_results(p1, &res[i], NULL)) { lz4amd_plan_destroy(p1); goto done; }
                lz4amd_plan_destroy(p1);
                if (res[i] < 0) { result = ERR(decompressionFailed); goto done; }
            }
            o += (size_t)res[i];
        }
        out_total = o;
    } else {
        for (i = 0; i < nb; i++) {
            d_dst[i] = (char*)g_stage.out + i * d->block_max;
            if (raw[i]) {
                if ((size_t)sizes[i] > d->block_max) { result = ERR(decompressionFailed); goto done; }
                res[i] = sizes[i];
            } else {
                d_src[ncomp] = (const char*)g_stage.in + in_off[i]; caps[ncomp] = (int)d->block_max;
                sums[ncomp] = sizes[i]; prefix[ncomp] = (int)i; ncomp++;
            }
        }
        if (ncomp) {
            void** dd = (void**)malloc(ncomp * sizeof *dd); int* rr = (int*)malloc(ncomp * sizeof *rr);
            if (!dd || !rr) { free(dd); free(rr); result = ERR(allocation_failed); goto done; }
            for (i = 0; i < ncomp; i++) dd[i] = d_dst[prefix[i]];
            if (lz4amd_plan_create(ctx, &dplan, LZ4AMD_OP_DECOMPRESS, (int)ncomp, d_src, sums, dd, caps, 0) ||
                lz4amd_plan_launch(dplan, NULL) || lz4amd_plan_results(dplan, rr, NULL)) { free(dd); free(rr); goto done; }
            for (i = 0; i < ncomp; i++) {
                if (rr[i] < 0) { free(dd); free(rr); result = ERR(decompressionFailed); goto done; }
                res[prefix[i]] = rr[i];
            }
            free(dd); free(rr);
        }
        for (i = 0; i < nb; i++) out_total += (size_t)res[i];
    }
    d->out = (uint8_t*)malloc(out_total ? out_total : 1);
    if (!d->out) { result = ERR(allocation_failed); goto done; }
    if (linked) { if (out_total && lz4amd_hip_d2h(d->out, g_stage.out, out_total, NULL)) goto done; }
    else {
        size_t o = 0;
        for (i = 0; i < nb; i++) {
            if (raw[i]) memcpy(d->out + o, base + in_off[i], (size_t)res[i]);
            else if (lz4amd_hip_d2h(d->out + o, d_dst[i], (size_t)res[i], NULL)) goto done;
            o += (size_t)res[i];
        }
    }
    if (lz4amd_hip_sync(NULL)) goto done;
    d->out_size = out_total;
    pthread_mutex_unlock(&lz4amd_default_lock);
    lz4amd_plan_destroy(dplan); lz4amd_plan_destroy(xplan);
    free(d_src); free(d_dst); free(sizes); free(caps); free(res); free(sums); free(prefix); free(in_off); free(raw);
checksum:
    if (d->info.contentSize && d->info.contentSize != d->out_size) return ERR(frameSize_wrong);     /* lz4frame.c:1984 */
    if (d->info.contentChecksumFlag && !skip_checksums) {
        const uint8_t* tail = d->in + d->frame_size - 4;
        if (rd32(tail) != xxh32(d->out, d->out_size)) return ERR(contentChecksum_invalid);          /* lz4frame.c:2021 */
    }
    d->decoded = 1;
    return 0;
done:
    pthread_mutex_unlock(&lz4amd_default_lock);
done_unlocked:
    lz4amd_plan_destroy(dplan); lz4amd_plan_destroy(xplan);
    free(d_src); free(d_dst); free(sizes); free(caps); free(res); free(sums); free(prefix); free(in_off); free(raw);
    return result;
}

size_t LZ4F_decompress(LZ4F_dctx* d, void* dstBuffer, size_t* dstSizePtr,
                       const void* srcBuffer, size_t* srcSizePtr, const LZ4F_decompressOptions_t* opt)
{
    const uint8_t* src = (const uint8_t*)srcBuffer;
    size_t avail, used = 0, dcap;
    if (!d || !dstSizePtr || !srcSizePtr) return ERR(parameter_null);
    avail = *srcSizePtr; dcap = *dstSizePtr;
    *srcSizePtr = 0; *dstSizePtr = 0;
    if (d->frame_done && d->decoded && d->out_pos >= d->out_size) LZ4F_resetDecompressionContext(d);   /* next frame */

    /* -- take input until the end of the frame is known and reached */
    while (!d->frame_done) {
        size_t want;                            /* bytes of the frame needed to make the next decision */
        if (!d->header_done) {
            size_t bm = 0, hs = parse_header(d->in, d->in_size, &d->info, &bm);
            if (LZ4F_isError(hs)) return hs;
            if (hs) {
                d->header_done = 1; d->header_size = hs; d->block_max = bm; d->scan_pos = hs;
                if (d->info.frameType == LZ4F_skippableFrame) { d->frame_size = hs + (size_t)d->info.contentSize; }
                continue;
            }
            if (d->in_size < 5) want = 5;
            else if ((rd32(d->in) & 0xFFFFFFF0u) == MAGIC_SKIP) want = 8;
            else want = 7 + ((d->in[4] & 8) ? 8 : 0) + ((d->in[4] & 1) ? 4 : 0);     /* never read past the header */
        } else if (d->info.frameType == LZ4F_skippableFrame) {
            if (d->in_size >= d->frame_size) { d->frame_done = 1; d->decoded = 1; d->out_size = d->out_pos = 0; break; }
            want = d->frame_size;
        } else if (d->in_size < d->scan_pos + 4) {
            want = d->scan_pos + 4;
        } else {
            const uint32_t f = rd32(d->in + d->scan_pos);
            if (f == 0) {                                           /* end mark */
                d->frame_size = d->scan_pos + 4 + (d->info.contentChecksumFlag ? 4 : 0);
                if (d->in_size >= d->frame_size) { d->frame_done = 1; break; }
                want = d->frame_size;
            } else {
                const size_t bsz = f & 0x7FFFFFFFu;
                if (bsz > d->block_max) return ERR(maxBlockSize_invalid);                  /* lz4frame.c:1737 */
                if (d->in_size >= d->scan_pos + 4 + bsz + (d->info.blockChecksumFlag ? 4 : 0)) {
                    d->scan_pos += 4 + bsz + (d->info.blockChecksumFlag ? 4 : 0);
                    continue;
                }
                want = d->scan_pos + 4 + bsz + (d->info.blockChecksumFlag ? 4 : 0);
            }
        }
        {   /* copy what is needed (and available) from the caller's buffer */
            size_t need = want - d->in_size, take = avail - used < need ? avail - used : need;
            if (take == 0) break;
            if (d->in_size + take > d->in_cap) {
                size_t nc = (d->in_size + take) * 2 + 4096;
                uint8_t* nbuf = (uint8_t*)realloc(d->in, nc);
                if (!nbuf) return ERR(allocation_failed);
                d->in = nbuf; d->in_cap = nc;
            }
            memcpy(d->in + d->in_size, src + used, take);
            d->in_size += take; used += take;
        }
    }
    *srcSizePtr = used;
    if (!d->frame_done) {                        /* hint: bytes still missing for the next step (>= 1) */
        return 4;
    }
    if (!d->decoded) {
        size_t r = decode_buffered_frame(d, opt && opt->skipChecksums);
        if (LZ4F_isError(r)) { LZ4F_resetDecompressionContext(d); return r; }
    }
    {   size_t left = d->out_size - d->out_pos, give = left < dcap ? left : dcap;
        if (give) memcpy(dstBuffer, d->out + d->out_pos, give);
        d->out_pos += give; *dstSizePtr = give;
        if (d->out_pos < d->out_size) return d->out_size - d->out_pos;       /* more output pending: call again */
    }
    return 0;
}
