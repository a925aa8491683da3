/*
 * lz4frame_api.c -- the LZ4 frame container (doc/lz4_Frame_format.md) around the GPU batch block
 * codec: one-shot LZ4F_compressFrame (lib/lz4frame.c:484 -> 428 -> compressBegin 690 / makeBlock
 * 883 / compressEnd 1206) and LZ4F_decompress (lz4frame.c:1613-2116; header 1346-1437).
 *
 * Host C only handles the container: magic, FLG/BD, optional content size / dictID, header
 * checksum, per-block size fields, end mark, content checksum.  LZ4F_compressFrame sends all blocks
 * of the frame to the device in ONE block table (lz4amd_batch.c); LZ4F_decompress is a streaming
 * state machine that decodes the complete blocks it holds as one table whenever the caller's input
 * runs dry, the frame ends or a batch (64 MiB of input / 1024 blocks / 256 MiB of output) is full, and delivers their bytes
 * before taking more input - memory is bounded by a batch, not by the frame.  The content checksum
 * is one serial XXH32 over the content (xxhash.c:352-389; the recurrence cannot be split), computed
 * on the calling thread.  There is no CPU codec here: a compressed block needs a HIP device (stored
 * blocks are copied, as in lz4frame.c:1790-1830), without one every call returns an error.
 */
#include "../../include/lz4frame.h"
#include "../../include/lz4amd.h"
#include "lz4amd_internal.h"
#include "lz4amd_ffi.h"
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
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
#include "xxh32_host.h"
static uint32_t xxh32(const uint8_t* p, size_t len) { return xxh32_once(p, len); }

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
/* Device staging and a stream per calling THREAD (the reference's frame functions are re-entrant; its CLI compresses and
 * decodes with -T4 workers): no lock is held while a frame's blocks are uploaded, coded and downloaded - the frames of
 * different threads overlap on the device.  (Rounds 1-3 kept one staging area behind the library's lock.) */
typedef struct { void* in; size_t in_cap; void* out; size_t out_cap; void* pack; size_t pack_cap; } dev_stage;
typedef struct { dev_stage st; void* stream; } frame_tls;
static pthread_key_t g_ftls_key;
static pthread_once_t g_ftls_once = PTHREAD_ONCE_INIT;
static void ftls_free(void* v)
{
    frame_tls* t = (frame_tls*)v;
    if (!t) return;
    lz4amd_hip_free(t->st.in); lz4amd_hip_free(t->st.out); lz4amd_hip_free(t->st.pack);
    lz4amd_hip_stream_destroy(t->stream);
    free(t);
}
static void ftls_key_init(void) { (void)pthread_key_create(&g_ftls_key, ftls_free); }
static frame_tls* ftls_get(void)
{   /* (after the default context exists: the stream belongs to its device) */
    frame_tls* t;
    pthread_once(&g_ftls_once, ftls_key_init);
    t = (frame_tls*)pthread_getspecific(g_ftls_key);
    if (!t) {
        t = (frame_tls*)calloc(1, sizeof *t);
        if (!t) return NULL;
        t->stream = lz4amd_hip_stream_create();
        if (!t->stream || pthread_setspecific(g_ftls_key, t)) { ftls_free(t); return NULL; }
    }
    return t;
}
/* Staging areas for work that runs BESIDE the calling thread (a decoder's batch on its helper thread): the thread's own area may
 * be used by whatever the caller does next on that thread - another frame function - so such work takes an area of its own,
 * from a small pool (device memory is slow to allocate; a decoder per frame then costs none). */
enum { kStagePool = 4 };
static frame_tls* g_stage_pool[kStagePool];
static pthread_mutex_t g_stage_lock = PTHREAD_MUTEX_INITIALIZER;
static frame_tls* stage_acquire(void)
{
    frame_tls* t = NULL;
    int i;
    pthread_mutex_lock(&g_stage_lock);
    for (i = 0; i < kStagePool && !t; i++) if (g_stage_pool[i]) { t = g_stage_pool[i]; g_stage_pool[i] = NULL; }
    pthread_mutex_unlock(&g_stage_lock);
    if (t) return t;
    t = (frame_tls*)calloc(1, sizeof *t);
    if (!t) return NULL;
    t->stream = lz4amd_hip_stream_create();
    if (!t->stream) { free(t); return NULL; }
    return t;
}
static void stage_release(frame_tls* t)
{
    int i;
    if (!t) return;
    pthread_mutex_lock(&g_stage_lock);
    for (i = 0; i < kStagePool; i++) if (!g_stage_pool[i]) { g_stage_pool[i] = t; t = NULL; break; }
    pthread_mutex_unlock(&g_stage_lock);
    ftls_free(t);                                  /* (the pool is full) */
}

static lz4amd_ctx* frame_ctx(void)
{
    lz4amd_ctx* c;
    pthread_mutex_lock(&lz4amd_default_lock);       /* (only the first call creates the context) */
    c = lz4amd_default_ctx();
    pthread_mutex_unlock(&lz4amd_default_lock);
    if (c && lz4amd_hip_use_device(c->device)) return NULL;
    return c;
}
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
/* The content checksum is one serial XXH32 over the source (lz4frame.c:1225): it runs on a helper thread from the
 * first byte uploaded to the last byte downloaded, instead of holding the calling thread for ~0.2 s per GiB. */
typedef struct { const uint8_t* p; size_t n; uint32_t h; } hash_job;
static void* hash_thread(void* arg) { hash_job* j = (hash_job*)arg; j->h = xxh32(j->p, j->n); return NULL; }

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
    frame_tls* ts; dev_stage* S; void* strm;
    lz4amd_plan *cplan = NULL, *xplan = NULL, *gplan = NULL;
    const void** d_src = NULL; void** d_dst = NULL; int *sizes = NULL, *caps = NULL, *csz = NULL, *sums = NULL, *pres = NULL;
    uint32_t content_sum = 0;
    int linked, hashing = 0;
    hash_job hj; pthread_t hthread;

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
    if (p.frameInfo.contentChecksumFlag) {
        hj.p = src; hj.n = srcSize; hj.h = 0;
        hashing = srcSize >= (1u << 20) && pthread_create(&hthread, NULL, hash_thread, &hj) == 0;
        if (!hashing) content_sum = xxh32(src, srcSize);
    }

    /* -- all blocks in one block table on the device */
    stride = (bs + bs / 255 + 16 + 255) & ~(size_t)255;
    d_src = (const void**)malloc(nb * sizeof *d_src); d_dst = (void**)malloc(nb * sizeof *d_dst);
    sizes = (int*)malloc(nb * sizeof *sizes); caps = (int*)malloc(nb * sizeof *caps);
    csz = (int*)malloc(nb * sizeof *csz); sums = (int*)malloc(nb * sizeof *sums); pres = (int*)malloc(nb * sizeof *pres);
    if (!d_src || !d_dst || !sizes || !caps || !csz || !sums || !pres) { result = ERR(allocation_failed); goto done_unlocked; }

    ctx = frame_ctx();
    ts = ctx ? ftls_get() : NULL;
    if (!ts) goto done;
    S = &ts->st; strm = ts->stream;
    if (stage_fit(&S->in, &S->in_cap, srcSize + 64) || stage_fit(&S->out, &S->out_cap, nb * stride)) { result = ERR(allocation_failed); goto done; }
    if (lz4amd_hip_h2d(S->in, src, srcSize, strm)) goto done;
    for (i = 0; i < nb; i++) {
        const size_t chunk = (i + 1 < nb) ? bs : srcSize - i * bs;
        d_src[i] = (const char*)S->in + i * bs; sizes[i] = (int)chunk;
        d_dst[i] = (char*)S->out + i * stride; caps[i] = (int)chunk - 1;      /* lz4frame.c:891-899: must gain a byte */
        if (caps[i] < 1) caps[i] = 1;
        pres[i] = linked ? (int)(i * bs < 65536 ? i * bs : 65536) : 0;     /* the history is the source itself */
    }
    if (p.compressionLevel >= 2) {
        /* lz4frame.c:943-958 LZ4F_selectCompression: levels >= LZ4HC_CLEVEL_MIN take the HC compressor */
        if (lz4amd_plan_create_compress_hc_prefix(ctx, &cplan, (int)nb, d_src, sizes, d_dst, caps, linked ? pres : NULL,
                                                  p.compressionLevel | (p.favorDecSpeed ? LZ4AMD_HC_FAVOR_DEC_SPEED : 0))) goto done;      /* lz4frame.c:713 */
    } else
    if (lz4amd_plan_create_compress_prefix(ctx, &cplan, (int)nb, d_src, sizes, d_dst, caps, linked ? pres : NULL)) goto done;
    if (p.compressionLevel < 0) (void)lz4amd_plan_set_acceleration(cplan, -p.compressionLevel + 1);      /* lz4frame.c:924-927: negative levels are accelerations */
    if (lz4amd_plan_launch(cplan, strm)) goto done;
    if (lz4amd_plan_results(cplan, csz, strm)) goto done;
    for (i = 0; i < nb; i++) if (csz[i] <= 0 || csz[i] >= sizes[i]) csz[i] = 0;   /* stored raw */
    if (p.frameInfo.blockChecksumFlag) {          /* lz4frame.c:904: XXH32 of the block as stored */
        for (i = 0; i < nb; i++) { if (csz[i]) { d_src[i] = d_dst[i]; caps[i] = csz[i]; } else caps[i] = sizes[i]; }
        if (lz4amd_plan_create(ctx, &xplan, LZ4AMD_OP_XXH32, (int)nb, d_src, caps, NULL, NULL, 0)) goto done;
        if (lz4amd_plan_launch(xplan, strm) || lz4amd_plan_results(xplan, sums, strm)) goto done;
    }
    {   /* The blocks sit in bound-sized slots; their sizes say where each goes in the frame (lz4frame.c:883-914 appends
         * them one behind the other).  One gather launch packs them on the device - stored blocks straight from the
         * source - and the body of the frame comes back in ONE transfer; the 4-byte size fields and checksums are
         * written into the gaps here. */
        const size_t tailsz = p.frameInfo.blockChecksumFlag ? 4 : 0;
        size_t total = 0;
        for (i = 0; i < nb; i++) {
            const size_t n = csz[i] ? (size_t)csz[i] : (size_t)sizes[i];
            d_src[i] = csz[i] ? (const char*)S->out + i * stride : (const char*)S->in + i * bs;
            caps[i] = (int)n;                                   /* rows: (source, size) -> body offset + 4 */
            pres[i] = (int)n;
            total += 4 + n + tailsz;
        }
        if (stage_fit(&S->pack, &S->pack_cap, total + 64)) { result = ERR(allocation_failed); goto done; }
        {   size_t off = 0;
            for (i = 0; i < nb; i++) { d_dst[i] = (char*)S->pack + off + 4; off += 4 + (size_t)caps[i] + tailsz; }
        }
        if (lz4amd_plan_create(ctx, &gplan, LZ4AMD_OP_GATHER, (int)nb, d_src, caps, d_dst, pres, 0) || lz4amd_plan_launch(gplan, strm)) goto done;
        if (lz4amd_hip_d2h(op, S->pack, total, strm) || lz4amd_hip_sync(strm)) goto done;
        for (i = 0; i < nb; i++) {
            const uint32_t n = (uint32_t)caps[i];
            wr32(op, csz[i] ? n : (n | 0x80000000u)); op += 4 + n;          /* lz4frame.c:896-903 */
            if (tailsz) { wr32(op, (uint32_t)sums[i]); op += 4; }           /* lz4frame.c:904-908 */
        }
    }
    if (lz4amd_hip_sync(strm)) goto done;
    lz4amd_plan_destroy(cplan); lz4amd_plan_destroy(xplan); lz4amd_plan_destroy(gplan); cplan = xplan = gplan = NULL;
    goto finish_frame_free;
done:
done_unlocked:
    if (hashing) pthread_join(hthread, NULL);
    lz4amd_plan_destroy(cplan); lz4amd_plan_destroy(xplan); lz4amd_plan_destroy(gplan);
    free(d_src); free(d_dst); free(sizes); free(caps); free(csz); free(sums); free(pres);
    return result;
finish_frame_free:
    if (hashing) { pthread_join(hthread, NULL); content_sum = hj.h; }
    free(d_src); free(d_dst); free(sizes); free(caps); free(csz); free(sums); free(pres);
finish_frame:
    if (nb == 0 && p.frameInfo.contentChecksumFlag) content_sum = xxh32(src, 0);
    wr32(op, 0); op += 4;                                                     /* end mark, lz4frame.c:1222 */
    if (p.frameInfo.contentChecksumFlag) { wr32(op, content_sum); op += 4; }   /* lz4frame.c:1225-1231 */
    return (size_t)(op - dst);
}

/* ---------------------------------------------------------------- decompression */
/* Streaming decoder (lz4frame.c:1613-2060 is the reference's state machine).  Input is copied into `in` one item at a
 * time (header, block header + block + block checksum, content checksum), never past the frame's end; complete
 * blocks are decoded in batches - as soon as the caller's input runs dry, the end mark shows up or kBatchBytes /
 * kBatchBlocks are buffered - and their bytes are delivered before more input is taken, so memory is bounded by
 * one batch whatever the frame's size and a flushed block never waits for the rest of the frame.  Linked blocks
 * (lz4frame.c:1901-1915) keep the last 64 KB of output on the host between batches. */
enum { ST_HEADER = 0, ST_SKIP, ST_BLOCKS, ST_TAIL, ST_DONE };
enum { kBatchBlocks = 1024 };
static const size_t kAsyncDecoded = (size_t)8 << 20;       /* a batch that may decode to this much runs on a helper thread, beside the caller's copies */
static const size_t kBatchBytes = (size_t)32 << 20;
static const size_t kBatchDecoded = (size_t)128 << 20;     /* decoded bytes a batch may need (its blocks x the frame's block size): a
                                                             * few MB of highly compressible 4 MiB blocks must not ask for GBs of memory */
struct LZ4F_dctx_s {
    int stage;
    uint8_t* in; size_t in_size, in_cap; int in_pin;   /* bytes of the item(s) being collected (_pin: page-locked memory) */
    uint8_t* in2; size_t in2_cap; int in2_pin;    /* the batch that is being decoded */
    uint8_t* out2; size_t out2_cap; int out2_pin; /* ... and where its bytes go (the two output buffers swap when it is done) */
    int out_pin;
    /* a large batch is decoded by a helper thread while the caller's thread hands out the batch before and takes in the next */
    int busy; pthread_t bthread; size_t b_nb, b_end, b_out_size, b_result; int b_skip; void* b_ts; int deferred;      /* deferred: the prepared batch (in2, b_nb, b_end) could not get a helper thread and waits for the bytes before it to be handed out */
     /* b_ts: the staging area of this context's helper-thread batches (from a pool) */
    size_t scan_pos; size_t nready;               /* complete blocks in in[0, scan_pos) */
    int end_seen;                                 /* in[scan_pos, scan_pos+4) is the end mark */
    uint8_t* out; size_t out_size, out_pos, out_cap;   /* decoded bytes of the last batch, and how many were delivered */
    LZ4F_frameInfo_t info;
    size_t block_max;
    unsigned long long total_out, skip_left;
    xxh32_state xxh;
    uint8_t* hist; size_t hist_len;               /* linked frames: the 64 KB before the next batch */
    LZ4F_CustomMem cmem; int has_cmem;            /* LZ4F_createDecompressionContext_advanced: who allocated the context */
    int skipc, skipc_call;                        /* skipChecksums: once asked for it holds for the rest of the frame (lz4frame.c:1634) */
    const uint8_t* dict; size_t dict_len;         /* LZ4F_decompress_usingDict: what the frame's first bytes (every block of an independent-block frame) may reference */
    int dict_keep;                                /* ... installed by the call that is running: the reset between two frames inside that call leaves it */
    /* the content checksum of a batch runs on a helper thread while its bytes are delivered and the next input is taken */
    int hashing; pthread_t hthread; size_t hash_n; const uint8_t* hash_p;
    /* a stored block that arrives in pieces goes straight from the caller's input to the caller's output (lz4frame.c:1790-1830) */
    int raw_on; size_t raw_left, raw_b0, raw_b1; xxh32_state raw_xxh;      /* raw_b0 .. raw_b1: bytes of the block that were buffered before (they go first) */
    int test_no_bthread;                          /* fault injection for the tests, read from the environment once, when the context is made (batch_thread_start) */
};
static void dctx_read_test_hooks(LZ4F_dctx* d) { const char* e = getenv("LZ4AMD_TEST_NO_BATCH_THREAD"); d->test_no_bthread = e && e[0] == '1'; }
static void* dctx_hash_thread(void* arg) { LZ4F_dctx* d = (LZ4F_dctx*)arg; xxh32_update(&d->xxh, d->hash_p, d->hash_n); return NULL; }
static void dctx_hash_join(LZ4F_dctx* d) { if (d->hashing) { pthread_join(d->hthread, NULL); d->hashing = 0; } }
static void dctx_batch_drop(LZ4F_dctx* d) { if (d->busy > 0) pthread_join(d->bthread, NULL); d->busy = 0; }      /* (a batch nobody waits for any more) */
/* Page-locked buffers are expensive to make (the pages are pinned one by one): the ones a context lets go of wait here for
 * the next context that needs one - a decoder per frame, as the reference's own programs create them, then costs no pinning. */
enum { kPinCache = 8 };
static const size_t kPinCacheBytes = (size_t)384 << 20;      /* page-locked bytes the cache may hold in all */
static struct { uint8_t* p; size_t cap; } g_pin_cache[kPinCache];
static pthread_mutex_t g_pin_lock = PTHREAD_MUTEX_INITIALIZER;
static uint8_t* pin_take(size_t need, size_t* cap)
{
    int i, best = -1;
    uint8_t* p = NULL;
    pthread_mutex_lock(&g_pin_lock);
    for (i = 0; i < kPinCache; i++) if (g_pin_cache[i].p && g_pin_cache[i].cap >= need && (best < 0 || g_pin_cache[i].cap < g_pin_cache[best].cap)) best = i;
    if (best >= 0) { p = g_pin_cache[best].p; *cap = g_pin_cache[best].cap; g_pin_cache[best].p = NULL; }
    pthread_mutex_unlock(&g_pin_lock);
    return p;
}
static void hfree_cap(uint8_t* p, int pinned, size_t cap)
{
    if (!p) return;
    if (pinned) {
        int i, slot = -1;
        pthread_mutex_lock(&g_pin_lock);
        size_t held = 0;
        for (i = 0; i < kPinCache; i++) if (g_pin_cache[i].p) held += g_pin_cache[i].cap;
        for (i = 0; i < kPinCache && held + cap <= kPinCacheBytes; i++) if (!g_pin_cache[i].p) { slot = i; break; }
        if (slot < 0 && held + cap <= kPinCacheBytes) for (i = 0; i < kPinCache; i++) if (g_pin_cache[i].cap < cap && (slot < 0 || g_pin_cache[i].cap < g_pin_cache[slot].cap)) slot = i;     /* (the smallest one makes room) */
        if (slot >= 0) { uint8_t* old = g_pin_cache[slot].p; g_pin_cache[slot].p = p; g_pin_cache[slot].cap = cap; p = old; }
        pthread_mutex_unlock(&g_pin_lock);
        lz4amd_hip_host_free(p);
    } else free(p);
}

LZ4F_errorCode_t LZ4F_createDecompressionContext(LZ4F_dctx** dctxPtr, unsigned version)
{   /* lz4frame.c:1284-1310 */
    if (!dctxPtr) return ERR(parameter_null);
    (void)version;
    *dctxPtr = (LZ4F_dctx*)calloc(1, sizeof **dctxPtr);
    if (*dctxPtr) dctx_read_test_hooks(*dctxPtr);
    return *dctxPtr ? 0 : ERR(allocation_failed);
}
void LZ4F_resetDecompressionContext(LZ4F_dctx* d)
{   /* lz4frame.c:1322-1330; the buffers are kept */
    if (!d) return;
    dctx_batch_drop(d);
    dctx_hash_join(d);
    d->stage = ST_HEADER;
    d->in_size = d->scan_pos = d->nready = 0; d->end_seen = 0; d->raw_on = 0; d->raw_left = 0; d->deferred = 0;
    d->out_size = d->out_pos = 0;
    d->total_out = d->skip_left = 0; d->hist_len = 0;
    if (!d->dict_keep) { d->dict = NULL; d->dict_len = 0; }      /* lz4frame.c:1331-1332: a dictionary counts for one frame */
}
LZ4F_errorCode_t LZ4F_freeDecompressionContext(LZ4F_dctx* d)
{
    if (d) {
        dctx_batch_drop(d);
        stage_release((frame_tls*)d->b_ts);
        dctx_hash_join(d); hfree_cap(d->in, d->in_pin, d->in_cap); hfree_cap(d->in2, d->in2_pin, d->in2_cap); hfree_cap(d->out, d->out_pin, d->out_cap); hfree_cap(d->out2, d->out2_pin, d->out2_cap); free(d->hist);
        if (d->has_cmem && d->cmem.customFree) d->cmem.customFree(d->cmem.opaqueState, d); else free(d);
    }
    return 0;
}
LZ4F_dctx* LZ4F_createDecompressionContext_advanced(LZ4F_CustomMem customMem, unsigned version)
{   /* lz4frame.c:1267-1279 (the context itself comes from the caller's allocator; the batch buffers are the library's) */
    LZ4F_dctx* d = NULL;
    (void)version;
    if (customMem.customCalloc) d = (LZ4F_dctx*)customMem.customCalloc(customMem.opaqueState, sizeof *d);
    else if (customMem.customAlloc) { d = (LZ4F_dctx*)customMem.customAlloc(customMem.opaqueState, sizeof *d); if (d) memset(d, 0, sizeof *d); }
    else d = (LZ4F_dctx*)calloc(1, sizeof *d);
    if (d) { d->cmem = customMem; d->has_cmem = customMem.customFree != NULL; dctx_read_test_hooks(d); }
    return d;
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

/* bytes of the header needed to know its size, then the size itself (lz4frame.c:1441-1462) */
static size_t header_want(const uint8_t* p, size_t n)
{
    if (n >= 4 && (rd32(p) & 0xFFFFFFF0u) == MAGIC_SKIP) return 8;
    if (n >= 4 && rd32(p) != MAGIC) return 4;    /* not a frame: parse_header says so at once (lz4frame.c:1375), before FLG is believed */
    if (n < 7) return 7;                         /* the smallest header; the smallest frame is 11 bytes, so this never reads past one */
    return 7 + ((p[4] & 8) ? 8 : 0) + ((p[4] & 1) ? 4 : 0);
}
static void seed_history(LZ4F_dctx* d)
{   /* lz4frame.c:1904-1912: a linked frame's first block sees the dictionary as the output before it */
    d->hist_len = 0;
    if (d->dict_len && d->info.frameType == LZ4F_frame && d->info.blockMode == LZ4F_blockLinked) {
        if (d->hist || (d->hist = (uint8_t*)malloc(65536))) { memcpy(d->hist, d->dict, d->dict_len); d->hist_len = d->dict_len; }
    }
}
static void enter_frame(LZ4F_dctx* d, size_t block_max)
{
    d->block_max = block_max;
    d->in_size = d->scan_pos = d->nready = 0; d->end_seen = 0;
    d->total_out = 0; d->hist_len = 0;
    d->skipc = d->skipc_call;                    /* a new frame: only what the current call asks for */
    seed_history(d);
    xxh32_reset(&d->xxh);
    if (d->info.frameType == LZ4F_skippableFrame) { d->skip_left = d->info.contentSize; d->stage = d->skip_left ? ST_SKIP : ST_DONE; }
    else d->stage = ST_BLOCKS;
}

size_t LZ4F_getFrameInfo(LZ4F_dctx* d, LZ4F_frameInfo_t* info, const void* srcBuffer, size_t* srcSizePtr)
{   /* lz4frame.c:1464-1512: decodes AND consumes the header; a header that is not whole is an error and consumes nothing */
    const uint8_t* src = (const uint8_t*)srcBuffer;
    size_t bm = 0, hs, avail;
    if (!d || !info || !srcSizePtr) return ERR(parameter_null);
    avail = *srcSizePtr; *srcSizePtr = 0;
    if (d->stage != ST_HEADER) { *info = d->info; return 4; }            /* lz4frame.c:1470-1477: already known */
    if (d->in_size) return ERR(frameDecoding_alreadyStarted);            /* lz4frame.c:1478-1482: in the middle of the header */
    if (!src || avail < 7) return ERR(frameHeader_incomplete);
    if (avail < header_want(src, avail)) return ERR(frameHeader_incomplete);
    hs = parse_header(src, avail, &d->info, &bm);
    if (LZ4F_isError(hs)) return hs;
    if (hs == 0) return ERR(frameHeader_incomplete);
    enter_frame(d, bm);
    *info = d->info; *srcSizePtr = hs;
    return 4;                                                            /* next: a block header */
}

/* The batch buffers: page-locked once they are large (the transfers then run at the bus's speed, and the device copies
 * straight out of / into them); small ones - headers, the frames of the unit tests - are ordinary memory. */
static int grow(uint8_t** buf, size_t* cap, int* pin, size_t need, size_t keep)
{
    uint8_t* nb = NULL;
    int np = 0;
    if (need <= *cap) return 0;
    if (need >= ((size_t)1 << 20)) {
        nb = pin_take(need, &need);
        if (nb) np = 1;
        else {
            need += (need >> 2) + 4096;                     /* (a quarter of slack: pinning pages is slow, but a context holds four such buffers) */
            if (frame_ctx()) { nb = (uint8_t*)lz4amd_hip_host_alloc(need); np = nb != NULL; }
        }
    } else need += (need >> 2) + 4096;
    if (!nb) nb = (uint8_t*)malloc(need);
    if (!nb) return -1;
    if (keep) memcpy(nb, *buf, keep);
    hfree_cap(*buf, *pin, *cap); *buf = nb; *cap = need; *pin = np;
    return 0;
}

/* linked frames: the 64 KB before whatever comes next (lz4frame.c:1901-1915) */
static int hist_append(LZ4F_dctx* d, const uint8_t* p, size_t n)
{
    if (!d->hist && !(d->hist = (uint8_t*)malloc(65536))) return -1;
    if (n >= 65536) { memcpy(d->hist, p + n - 65536, 65536); d->hist_len = 65536; }
    else {
        const size_t keep = d->hist_len + n > 65536 ? 65536 - n : d->hist_len;
        memmove(d->hist, d->hist + d->hist_len - keep, keep);
        memcpy(d->hist + keep, p, n);
        d->hist_len = keep + n;
    }
    return 0;
}

/* decode the nb complete blocks held in base[0, end) (d->in2) into d->out2; d->b_out_size: how many bytes that made.
 * Runs on the caller's thread or on the batch's helper thread (ts: the staging area and stream of the thread that owns the context). */
static size_t decode_batch(LZ4F_dctx* d, const uint8_t* const base, size_t nb, size_t end, int skip_checksums, frame_tls* ts_given)
{
    size_t pos, i, out_total = 0, result = ERR(GENERIC);
    const int bchk = d->info.blockChecksumFlag == LZ4F_blockChecksumEnabled;
    const int linked = d->info.blockMode == LZ4F_blockLinked;
    const size_t h0 = linked ? d->hist_len : 0;
    lz4amd_ctx* ctx;
    frame_tls* ts = NULL; dev_stage* S; void* strm = NULL;
    lz4amd_plan *dplan = NULL, *xplan = NULL;
    const void** d_src = NULL; void** d_dst = NULL; int *sizes = NULL, *caps = NULL, *res = NULL, *sums = NULL, *prefix = NULL;
    size_t* in_off = NULL; uint8_t* raw = NULL;
    size_t ncomp = 0;

    d->b_out_size = 0;
    d_src = (const void**)malloc(nb * sizeof *d_src); d_dst = (void**)malloc(nb * sizeof *d_dst);
    sizes = (int*)malloc(nb * sizeof *sizes); caps = (int*)malloc(nb * sizeof *caps); res = (int*)malloc(nb * sizeof *res);
    sums = (int*)malloc(nb * sizeof *sums); prefix = (int*)malloc(nb * sizeof *prefix);
    in_off = (size_t*)malloc(nb * sizeof *in_off); raw = (uint8_t*)malloc(nb);
    if (!d_src || !d_dst || !sizes || !caps || !res || !sums || !prefix || !in_off || !raw) { result = ERR(allocation_failed); goto done_unlocked; }
    for (pos = 0, i = 0; i < nb; i++) {
        const uint32_t f = rd32(base + pos);
        sizes[i] = (int)(f & 0x7FFFFFFFu); raw[i] = (uint8_t)(f >> 31); in_off[i] = pos + 4;
        pos += 4 + (size_t)sizes[i] + (bchk ? 4 : 0);
    }
    if (pos != end) goto done_unlocked;
    {   /* a batch of stored blocks only (lz4frame.c:1790-1830 copies them too) never visits the device */
        int all_raw = 1;
        for (i = 0; i < nb && all_raw; i++) if (!raw[i]) all_raw = 0;
        if (all_raw) {
            size_t o = 0;
            if (bchk && !skip_checksums)            /* (their checksums on the host as well: the bytes are copied here anyway; lz4frame.c:1878) */
                for (i = 0; i < nb; i++) if (xxh32(base + in_off[i], (size_t)sizes[i]) != rd32(base + in_off[i] + (size_t)sizes[i])) { result = ERR(blockChecksum_invalid); goto done_unlocked; }
            for (i = 0; i < nb; i++) { if ((size_t)sizes[i] > d->block_max) { result = ERR(decompressionFailed); goto done_unlocked; } out_total += (size_t)sizes[i]; }
            if (grow(&d->out2, &d->out2_cap, &d->out2_pin, out_total ? out_total : 1, 0)) { result = ERR(allocation_failed); goto done_unlocked; }
            for (i = 0; i < nb; i++) { memcpy(d->out2 + o, base + in_off[i], (size_t)sizes[i]); o += (size_t)sizes[i]; }
            goto host_tail;
        }
    }

    ctx = frame_ctx();
    ts = ctx ? (ts_given ? ts_given : ftls_get()) : NULL;
    if (!ts) goto done;
    S = &ts->st; strm = ts->stream;
    if (stage_fit(&S->in, &S->in_cap, end + 64) || stage_fit(&S->out, &S->out_cap, h0 + nb * d->block_max + 64)) { result = ERR(allocation_failed); goto done; }
    if (lz4amd_hip_h2d(S->in, base, end, strm)) goto done;
    /* block checksums: XXH32 of every block as stored (lz4frame.c:1878), one launch */
    if (bchk && !skip_checksums) {
        for (i = 0; i < nb; i++) d_src[i] = (const char*)S->in + in_off[i];
        if (lz4amd_plan_create(ctx, &xplan, LZ4AMD_OP_XXH32, (int)nb, d_src, sizes, NULL, NULL, 0)) goto done;
        if (lz4amd_plan_launch(xplan, strm) || lz4amd_plan_results(xplan, sums, strm)) goto done;
        for (i = 0; i < nb; i++) if ((uint32_t)sums[i] != rd32(base + in_off[i] + (size_t)sizes[i])) { result = ERR(blockChecksum_invalid); goto done; }
    }
    if (linked) {
        /* lz4frame.c:1901-1915: the previous 64 KB of output are the dictionary.  The history of the earlier batches
         * goes in front of the packed output; the blocks are decoded in order, each told how much precedes it. */
        size_t o = h0;
        unsigned char* st = (unsigned char*)malloc(nb);
        if (!st) { result = ERR(allocation_failed); goto done; }
        if (h0 && lz4amd_hip_h2d(S->out, d->hist, h0, strm)) { free(st); goto done; }
        for (i = 0; i < nb; i++) {
            d_src[i] = (const char*)S->in + in_off[i]; caps[i] = (int)d->block_max; st[i] = raw[i];
            if (raw[i] && (size_t)sizes[i] > d->block_max) { free(st); result = ERR(decompressionFailed); goto done; }
        }
        /* one launch for the whole batch: every workgroup pre-parses its block at once, only the copy stages run one after
         * the other (lz4amd_plan_create_decompress_chained); stored blocks are copied in place by the same launch */
        if (lz4amd_plan_create_decompress_chained(ctx, &dplan, (int)nb, d_src, sizes, (char*)S->out + h0, caps, st, (int)h0)
            || lz4amd_plan_launch(dplan, strm) || lz4amd_plan_results(dplan, res, strm)) { free(st); goto done; }
        free(st);
        for (i = 0; i < nb; i++) {
            if (res[i] < 0) {
                result = ERR(decompressionFailed); goto done;
            }
            o += (size_t)res[i];
        }
        out_total = o - h0;
    } else {
        const size_t dl = d->dict_len, slot = d->block_max + (dl ? 65536 : 0);
        if (dl) {
            /* LZ4F_decompress_usingDict on independent blocks (lz4frame.c:1904): the dictionary goes in front of every slot
             * (one upload, one gather launch), the decoder's prefix mode does the rest */
            lz4amd_plan* cp = NULL;
            int* dls = (int*)malloc(nb * sizeof *dls);
            int ok = dls != NULL && !stage_fit(&S->out, &S->out_cap, nb * slot + 64) && !stage_fit(&S->pack, &S->pack_cap, dl + 64)
                     && !lz4amd_hip_h2d(S->pack, d->dict, dl, strm);
            for (i = 0; ok && i < nb; i++) { d_src[i] = S->pack; dls[i] = (int)dl; d_dst[i] = (char*)S->out + i * slot + (65536 - dl); }
            ok = ok && !lz4amd_plan_create(ctx, &cp, LZ4AMD_OP_GATHER, (int)nb, d_src, dls, d_dst, dls, 0) && !lz4amd_plan_launch(cp, strm);
            lz4amd_plan_destroy(cp); free(dls);
            if (!ok) { result = ERR(allocation_failed); goto done; }
        }
        for (i = 0; i < nb; i++) {
            d_dst[i] = (char*)S->out + i * slot + (dl ? 65536 : 0);
            if (raw[i]) {
                if ((size_t)sizes[i] > d->block_max) { result = ERR(decompressionFailed); goto done; }
                res[i] = sizes[i];
            } else {
                d_src[ncomp] = (const char*)S->in + in_off[i]; caps[ncomp] = (int)d->block_max;
                sums[ncomp] = sizes[i]; prefix[ncomp] = (int)i; ncomp++;
            }
        }
        if (ncomp) {
            void** dd = (void**)malloc(ncomp * sizeof *dd); int* rr = (int*)malloc(ncomp * sizeof *rr);
            if (!dd || !rr) { free(dd); free(rr); result = ERR(allocation_failed); goto done; }
            for (i = 0; i < ncomp; i++) dd[i] = d_dst[prefix[i]];
            for (i = 0; dl && i < ncomp; i++) rr[i] = (int)dl;              /* (prefix lengths; rr[] takes the results afterwards) */
            if ((dl ? lz4amd_plan_create_prefix(ctx, &dplan, (int)ncomp, d_src, sums, dd, caps, rr)
                    : lz4amd_plan_create(ctx, &dplan, LZ4AMD_OP_DECOMPRESS, (int)ncomp, d_src, sums, dd, caps, 0)) ||
                lz4amd_plan_launch(dplan, strm) || lz4amd_plan_results(dplan, rr, strm)) { free(dd); free(rr); goto done; }
            for (i = 0; i < ncomp; i++) {
                if (rr[i] < 0) { free(dd); free(rr); result = ERR(decompressionFailed); goto done; }
                res[prefix[i]] = rr[i];
            }
            free(dd); free(rr);
        }
        for (i = 0; i < nb; i++) out_total += (size_t)res[i];
    }
    if (grow(&d->out2, &d->out2_cap, &d->out2_pin, out_total ? out_total : 1, 0)) { result = ERR(allocation_failed); goto done; }
    if (linked) { if (out_total && lz4amd_hip_d2h(d->out2, (char*)S->out + h0, out_total, strm)) goto done; }
    else {
        /* every block but the last is normally full, so the block_max-strided slots ARE the content: one transfer
         * (stored blocks are then laid over their slots from the input); ragged tables go block by block */
        size_t o = 0;
        int dense = 1;
        if (d->dict_len) dense = 0;                  /* (the slots are 64 KB apart then) */
        for (i = 0; i + 1 < nb; i++) if ((size_t)res[i] != d->block_max) { dense = 0; break; }
        if (dense && out_total && lz4amd_hip_d2h(d->out2, S->out, out_total, strm)) goto done;
        for (i = 0; i < nb; i++) {
            if (!raw[i] && !dense && lz4amd_hip_d2h(d->out2 + o, d_dst[i], (size_t)res[i], strm)) goto done;
            o += (size_t)res[i];
        }
    }
    if (lz4amd_hip_sync(strm)) goto done;
    if (!linked) {          /* (after the transfers have landed: the buffer is page-locked memory, a copy into it is still on its way when the call returns) */
        size_t o = 0;
        for (i = 0; i < nb; i++) { if (raw[i]) memcpy(d->out2 + o, base + in_off[i], (size_t)res[i]); o += (size_t)res[i]; }
    }
host_tail:
    d->b_out_size = out_total; d->total_out += out_total;
    if (d->info.contentChecksumFlag && !skip_checksums) {                   /* lz4frame.c:1896, 1967 */
        dctx_hash_join(d);                          /* (the batch before: one serial hash, batch after batch; its buffer is the one written next) */
        d->hash_n = out_total; d->hash_p = d->out2;
        d->hashing = out_total >= (1u << 20) && pthread_create(&d->hthread, NULL, dctx_hash_thread, d) == 0;
        if (!d->hashing) xxh32_update(&d->xxh, d->out2, out_total);
    }
    if (linked && hist_append(d, d->out2, out_total)) { result = ERR(allocation_failed); goto done_unlocked; }      /* the 64 KB the next batch may reference */
    result = 0;
    goto done_unlocked;
done:
    if (ts && strm) (void)lz4amd_hip_sync(strm);     /* (nothing may still be on its way into or out of the page-locked buffers) */
done_unlocked:
    lz4amd_plan_destroy(dplan); lz4amd_plan_destroy(xplan);
    free(d_src); free(d_dst); free(sizes); free(caps); free(res); free(sums); free(prefix); free(in_off); free(raw);
    return result;
}

/* append up to `want - in_size` bytes of the caller's input to d->in; returns 1 when in_size reached want */
static int take_input(LZ4F_dctx* d, size_t want, const uint8_t* src, size_t avail, size_t* used, size_t* err)
{
    size_t need, take;
    if (d->in_size >= want) return 1;
    need = want - d->in_size; take = avail - *used < need ? avail - *used : need;
    if (take) {
        if (grow(&d->in, &d->in_cap, &d->in_pin, d->in_size + take, d->in_size)) { *err = ERR(allocation_failed); return 0; }
        memcpy(d->in + d->in_size, src + *used, take);
        d->in_size += take; *used += take;
    }
    return d->in_size >= want;
}

/* bytes the decoder would like to see next (lz4frame.c's nextSrcSizeHint): what the current item still
 * misses, plus the next block header when the item is a block */
static size_t size_hint(const LZ4F_dctx* d)
{
    const size_t tail = d->info.blockChecksumFlag == LZ4F_blockChecksumEnabled ? 4 : 0;
    switch (d->stage) {
    case ST_HEADER: { const size_t w = header_want(d->in, d->in_size); return (w > d->in_size ? w - d->in_size : 0) + 4; }   /* lz4frame.c:1691, 1710: + the first block header */
    case ST_SKIP:   return d->skip_left > ((size_t)1 << 30) ? ((size_t)1 << 30) : (size_t)d->skip_left;
    case ST_BLOCKS:
        if (d->raw_on) return d->raw_left ? d->raw_left + tail + 4 : (tail - d->in_size) + 4;      /* lz4frame.c:1826: what the stored block misses + the next header */
        if (d->end_seen) return d->info.contentChecksumFlag ? 4 : 0;
        if (d->in_size < d->scan_pos + 4) return d->scan_pos + 4 - d->in_size;
        {   const size_t bsz = rd32(d->in + d->scan_pos) & 0x7FFFFFFFu;
            const size_t full = d->scan_pos + 4 + bsz + tail;
            return (full > d->in_size ? full - d->in_size : 0) + 4; }
    case ST_TAIL:   return 4 - d->in_size;
    default:        return 0;
    }
}

/* (test hook: LZ4AMD_TEST_NO_BATCH_THREAD=1 in the environment WHEN THE CONTEXT IS CREATED makes the helper thread of a large batch
 *  fail to start, the way pthread_create does when the process is out of threads; the environment is not looked at again) */
static int batch_thread_start(LZ4F_dctx* d, void* (*fn)(void*))
{
    if (d->test_no_bthread) return -1;
    return pthread_create(&d->bthread, NULL, fn, d);
}
static void* batch_thread(void* arg)
{
    LZ4F_dctx* d = (LZ4F_dctx*)arg;
    d->b_result = decode_batch(d, d->in2, d->b_nb, d->b_end, d->b_skip, (frame_tls*)d->b_ts);
    return NULL;
}
/* the batch in flight is done: its bytes become the ones to hand out (every byte of the batch before has been handed out) */
static size_t batch_finish(LZ4F_dctx* d)
{
    if (d->busy > 0) pthread_join(d->bthread, NULL);
    d->busy = 0;
    if (LZ4F_isError(d->b_result)) return d->b_result;
    { uint8_t* t = d->out; const size_t c = d->out_cap; const int pn = d->out_pin;
      d->out = d->out2; d->out_cap = d->out2_cap; d->out_pin = d->out2_pin; d->out2 = t; d->out2_cap = c; d->out2_pin = pn; }
    d->out_size = d->b_out_size; d->out_pos = 0;
    return 0;
}

size_t LZ4F_decompress(LZ4F_dctx* d, void* dstBuffer, size_t* dstSizePtr,
                       const void* srcBuffer, size_t* srcSizePtr, const LZ4F_decompressOptions_t* opt)
{
    const uint8_t* src = (const uint8_t*)srcBuffer;
    uint8_t* dst = (uint8_t*)dstBuffer;
    size_t avail, used = 0, dcap, given = 0, err = 0;
    if (!d || !dstSizePtr || !srcSizePtr) return ERR(parameter_null);
    d->skipc_call = opt && opt->skipChecksums;
    d->skipc |= d->skipc_call;
    avail = src ? *srcSizePtr : 0; dcap = dst ? *dstSizePtr : 0;
    *srcSizePtr = 0; *dstSizePtr = 0;

    for (;;) {
        /* -- decoded bytes go out before anything else is taken in */
        if (d->out_pos < d->out_size) {
            const size_t left = d->out_size - d->out_pos, give = left < dcap - given ? left : dcap - given;
            if (give) memcpy(dst + given, d->out + d->out_pos, give);
            d->out_pos += give; given += give;
            if (d->out_pos < d->out_size) break;                            /* dst is full: call again */
        }
        if (d->stage == ST_DONE) {                                           /* frame complete and delivered */
            LZ4F_resetDecompressionContext(d);
            *srcSizePtr = used; *dstSizePtr = given;
            return 0;
        }
        if (d->stage == ST_HEADER) {
            size_t bm = 0, hs;
            if (!take_input(d, header_want(d->in, d->in_size), src, avail, &used, &err)) { if (err) goto fail; break; }
            if (d->in_size < header_want(d->in, d->in_size)) continue;       /* the size is known now: take the rest */
            hs = parse_header(d->in, d->in_size, &d->info, &bm);
            if (LZ4F_isError(hs)) { err = hs; goto fail; }
            if (hs == 0) { err = ERR(frameHeader_incomplete); goto fail; }
            enter_frame(d, bm);
            continue;
        }
        if (d->stage == ST_SKIP) {                                           /* lz4frame.c:2037-2055: user data is skipped, not kept */
            const unsigned long long n = avail - used < d->skip_left ? avail - used : d->skip_left;
            used += (size_t)n; d->skip_left -= n;
            if (d->skip_left) break;
            d->stage = ST_DONE;
            continue;
        }
        if (d->stage == ST_TAIL) {                                           /* content checksum, lz4frame.c:2005-2030 */
            if (!take_input(d, 4, src, avail, &used, &err)) { if (err) goto fail; break; }
            dctx_hash_join(d);
            if (!d->skipc && rd32(d->in) != xxh32_digest(&d->xxh)) { err = ERR(contentChecksum_invalid); goto fail; }
            d->in_size = 0; d->stage = ST_DONE;
            continue;
        }
        /* -- ST_BLOCKS: collect whole blocks, decode them when there is no reason to wait for more */
        {   const size_t tail = d->info.blockChecksumFlag == LZ4F_blockChecksumEnabled ? 4 : 0;
            int starved = 0;
            if (d->raw_on) {
                /* a stored block in pieces: from the caller's input straight to the caller's output, as far as both reach */
                while (d->raw_left) {
                    const int buffered = d->raw_b0 < d->raw_b1;
                    const uint8_t* from = buffered ? d->in + d->raw_b0 : src + used;
                    size_t n = d->raw_left;
                    const size_t have = buffered ? d->raw_b1 - d->raw_b0 : avail - used;
                    if (n > have) n = have;
                    if (n > dcap - given) n = dcap - given;
                    if (!n) break;
                    if (d->info.contentChecksumFlag && !d->skipc) xxh32_update(&d->xxh, from, n);
                    if (tail && !d->skipc) xxh32_update(&d->raw_xxh, from, n);
                    if (d->info.blockMode == LZ4F_blockLinked && hist_append(d, from, n)) { err = ERR(allocation_failed); goto fail; }
                    memcpy(dst + given, from, n);
                    if (buffered) {
                        d->raw_b0 += n;
                        if (d->raw_b0 == d->raw_b1) {                        /* (what was buffered behind the block's bytes - a piece of its checksum - stays) */
                            memmove(d->in, d->in + d->raw_b1, d->in_size - d->raw_b1);
                            d->in_size -= d->raw_b1; d->raw_b0 = d->raw_b1 = 0;
                        }
                    } else used += n;
                    given += n; d->raw_left -= n; d->total_out += n;
                }
                if (d->raw_left) break;                                      /* more input, or more room */
                if (tail) {
                    if (!take_input(d, 4, src, avail, &used, &err)) { if (err) goto fail; break; }
                    if (!d->skipc && rd32(d->in) != xxh32_digest(&d->raw_xxh)) { err = ERR(blockChecksum_invalid); goto fail; }   /* lz4frame.c:1878 */
                    d->in_size = 0;
                }
                d->raw_on = 0;
                continue;
            }
            while (!d->end_seen && d->nready < (size_t)kBatchBlocks && d->scan_pos < kBatchBytes && (d->nready + 1) * d->block_max <= kBatchDecoded) {
                uint32_t f;
                if (!take_input(d, d->scan_pos + 4, src, avail, &used, &err)) { if (err) goto fail; starved = 1; break; }
                f = rd32(d->in + d->scan_pos);
                if (f == 0) { d->end_seen = 1; break; }                      /* end mark, lz4frame.c:1730 */
                if ((f & 0x7FFFFFFFu) > d->block_max) { err = ERR(maxBlockSize_invalid); goto fail; }   /* lz4frame.c:1737 */
                if ((f >> 31) && d->nready == 0 && !d->busy && !d->deferred && (size_t)(f & 0x7FFFFFFFu) + tail > avail - used) {
                    /* a stored block that is not all here, with nothing before it waiting for the device: its bytes need no buffer
                     * (lz4frame.c:1790-1830 hands them on as they come; a whole one at hand goes with the batch, one copy either way) */
                    dctx_hash_join(d);
                    d->raw_on = 1; d->raw_left = f & 0x7FFFFFFFu;
                    d->raw_b0 = 4; d->raw_b1 = d->in_size < 4 + d->raw_left ? d->in_size : 4 + d->raw_left;      /* (bytes an earlier call left in the buffer) */
                    if (d->raw_b0 == d->raw_b1) {                             /* nothing of the block's bytes is buffered - but a piece of its checksum may be (an empty stored block) */
                        memmove(d->in, d->in + 4, d->in_size - 4);
                        d->in_size -= 4; d->raw_b0 = d->raw_b1 = 0;
                    }
                    if (tail) xxh32_reset(&d->raw_xxh);
                    break;
                }
                if (!take_input(d, d->scan_pos + 4 + (f & 0x7FFFFFFFu) + tail, src, avail, &used, &err)) { if (err) goto fail; starved = 1; break; }
                d->scan_pos += 4 + (f & 0x7FFFFFFFu) + tail; d->nready++;
            }
            if (d->raw_on) continue;                                         /* (a stored block in pieces: handled at the top) */
            if (d->deferred) {
                /* a prepared batch whose helper thread could not be started: it is decoded here, once every byte before it has been
                 * handed out (decoding it now would put its bytes where those still wait) */
                size_t r;
                if (d->busy) { r = batch_finish(d); if (LZ4F_isError(r)) { err = r; goto fail; } }
                if (d->out_pos < d->out_size) continue;
                d->deferred = 0;
                d->b_result = decode_batch(d, d->in2, d->b_nb, d->b_end, d->b_skip, NULL);
                d->busy = -1;
                r = batch_finish(d);
                if (LZ4F_isError(r)) { err = r; goto fail; }
                continue;
            }
            if (d->nready) {                                                 /* starved, end of frame or a full batch */
                /* The batch goes to the device - a large one on a helper thread, so that this thread hands out the batch
                 * before it and takes in the one after it meanwhile (its input and output buffers are the spare pair). */
                const int large = d->nready * d->block_max >= kAsyncDecoded ;
                int can_async;
                size_t rest, r;
                if (d->busy) { r = batch_finish(d); if (LZ4F_isError(r)) { err = r; goto fail; } }
                can_async = large && frame_ctx() && (d->b_ts || (d->b_ts = stage_acquire()) != NULL);
                if (!can_async && d->out_pos < d->out_size) continue;         /* (its bytes would take this buffer: the pending ones go out first) */
                rest = d->in_size - d->scan_pos;                              /* the item that is still incomplete stays with the collector */
                if (grow(&d->in2, &d->in2_cap, &d->in2_pin, rest ? rest : 1, 0)) { err = ERR(allocation_failed); goto fail; }
                memcpy(d->in2, d->in + d->scan_pos, rest);
                { uint8_t* t = d->in; const size_t c = d->in_cap; const int pn = d->in_pin;
                  d->in = d->in2; d->in_cap = d->in2_cap; d->in_pin = d->in2_pin; d->in2 = t; d->in2_cap = c; d->in2_pin = pn; }
                d->b_nb = d->nready; d->b_end = d->scan_pos; d->b_skip = d->skipc;
                d->in_size = rest; d->scan_pos = 0; d->nready = 0;
                if (can_async && batch_thread_start(d, batch_thread) == 0) d->busy = 1;
                else if (d->out_pos < d->out_size) d->deferred = 1;           /* (no thread to be had, and the batch before is not handed out yet: see above) */
                else {
                    d->b_result = decode_batch(d, d->in2, d->b_nb, d->b_end, d->b_skip, NULL);
                    d->busy = -1;                                             /* (done already: nothing to join) */
                    r = batch_finish(d);
                    if (LZ4F_isError(r)) { err = r; goto fail; }
                }
                continue;
            }
            if (d->busy) {                                                   /* nothing else to do before its bytes are there */
                const size_t r = batch_finish(d);
                if (LZ4F_isError(r)) { err = r; goto fail; }
                continue;
            }
            if (d->end_seen) {
                if (d->info.contentSize && d->info.contentSize != d->total_out) { err = ERR(frameSize_wrong); goto fail; }   /* lz4frame.c:1984 */
                d->in_size = 0; d->end_seen = 0;
                d->stage = d->info.contentChecksumFlag ? ST_TAIL : ST_DONE;
                continue;
            }
            (void)starved;
            break;                                                           /* needs more input */
        }
    }
    *srcSizePtr = used; *dstSizePtr = given;
    {   const size_t h = size_hint(d);
        return h ? h : 1;                                                    /* output pending with nothing more to read: call again */
    }
fail:
    LZ4F_resetDecompressionContext(d);                                       /* lz4frame.c:2034: an error leaves the context reusable */
    *srcSizePtr = used; *dstSizePtr = given;
    return err;
}


/* ---------------------------------------------------------------- the frame API's long tail (what tests/frametest.c links) */
size_t LZ4F_getBlockSize(LZ4F_blockSizeID_t blockSizeID)
{   /* lz4frame.c:333-341 */
    const size_t bs = block_size_of((unsigned)blockSizeID);
    return bs ? bs : ERR(maxBlockSize_invalid);
}
size_t LZ4F_headerSize(const void* src, size_t srcSize)
{   /* lz4frame.c:1441-1462 */
    const uint8_t* p = (const uint8_t*)src;
    if (src == NULL) return ERR(srcPtr_wrong);
    if (srcSize < 5) return ERR(frameHeader_incomplete);                    /* LZ4F_MIN_SIZE_TO_KNOW_HEADER_LENGTH */
    if ((rd32(p) & 0xFFFFFFF0u) == MAGIC_SKIP) return 8;
    if (rd32(p) != MAGIC) return ERR(frameType_unknown);
    return 7 + ((p[4] & 8) ? 8 : 0) + ((p[4] & 1) ? 4 : 0);
}
size_t LZ4F_decompress_usingDict(LZ4F_dctx* dctx, void* dstBuffer, size_t* dstSizePtr, const void* srcBuffer, size_t* srcSizePtr,
                                 const void* dict, size_t dictSize, const LZ4F_decompressOptions_t* decompressOptionsPtr)
{   /* lz4frame.c:2070-2083: the dictionary counts when a frame starts; it must stay in place while the frame is decoded */
    size_t r;
    /* the header was consumed (LZ4F_getFrameInfo, to read the dictID) and nothing of the first block yet: the reference
     * still takes the dictionary there (dStage <= dstage_init) */
    int fresh;
    if (dctx == NULL) return ERR(parameter_null);
    if (dctx->stage == ST_DONE && dctx->out_pos >= dctx->out_size) LZ4F_resetDecompressionContext(dctx);      /* the frame before is complete and delivered */
    fresh = dctx->stage == ST_BLOCKS && !dctx->busy && dctx->total_out == 0 && dctx->in_size == 0 && dctx->nready == 0 && dctx->out_size == 0 && !dctx->end_seen;
    if (dctx->stage == ST_HEADER || dctx->stage == ST_DONE || fresh) {
        if (dict && dictSize) {
            if (dictSize > 65536) { dict = (const uint8_t*)dict + (dictSize - 65536); dictSize = 65536; }
            dctx->dict = (const uint8_t*)dict; dctx->dict_len = dictSize;
        } else { dctx->dict = NULL; dctx->dict_len = 0; }
        if (fresh) seed_history(dctx);
        if (dctx->stage == ST_DONE) dctx->dict_keep = 1;          /* (bytes of the frame before are still to be delivered: the reset comes inside this call) */
    }
    r = LZ4F_decompress(dctx, dstBuffer, dstSizePtr, srcBuffer, srcSizePtr, decompressOptionsPtr);
    dctx->dict_keep = 0;
    return r;
}
