/*
 * lz4_api.c -- the reference's classic one-block C ABI (lib/lz4.h) on top of the GPU batch codec.
 *
 * Same names, argument meaning and return conventions as the reference so existing callers can
 * relink: host pointers in, host pointers out; the block makes a round trip through HBM
 * (upload, kernels, download).  A lone small block cannot amortise that - the batch API in
 * lz4amd.h is what the benchmarks use - but the semantics are identical, which is what the
 * parity tests check.  There is deliberately NO CPU codec in this library: without a usable
 * HIP device compress returns 0 and decompress returns a negative value, after a message on
 * stderr.
 */
#include "../../include/lz4.h"
#include "../../include/lz4amd.h"
#include "lz4amd_internal.h"
#include "lz4amd_ffi.h"
#include <pthread.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

pthread_mutex_t lz4amd_default_lock = PTHREAD_MUTEX_INITIALIZER;   /* shared with lz4frame_api.c */
#define g_lock lz4amd_default_lock
static lz4amd_ctx* g_ctx = NULL;
static int g_ctx_failed = 0;

lz4amd_ctx* lz4amd_default_ctx(void)
{   /* caller holds g_lock */
    if (!g_ctx && !g_ctx_failed) {
        const char* e = getenv("LZ4AMD_DEVICE");
        if (lz4amd_ctx_create(&g_ctx, e ? atoi(e) : 0) != LZ4AMD_OK) g_ctx_failed = 1;
    }
    return g_ctx;
}

static int stage_reserve(void** buf, size_t* cap, size_t need)
{
    if (need <= *cap) return 0;
    lz4amd_hip_free(*buf);
    *cap = 0;
    *buf = lz4amd_hip_malloc(need + (need >> 2) + 4096);
    if (!*buf) return -1;
    *cap = need + (need >> 2) + 4096;
    return 0;
}

int LZ4_versionNumber(void) { return LZ4_VERSION_NUMBER; }
const char* LZ4_versionString(void) { return LZ4_VERSION_STRING; }
int LZ4_compressBound(int inputSize) { return lz4amd_compress_bound(inputSize); }

/* One block through the device; op selects the kernel.  Returns the per-block result, or `fail` when the device path
 * cannot run.  The reference's block functions are re-entrant and its CLI calls them from up to 200 threads
 * (lz4conf.h:60-61): every calling thread keeps its own stream, device staging buffers and one-row plans (no device
 * allocation, no lock in the steady state; the blocks of different threads run side by side on different CUs). */
typedef struct {
    void* stream;
    void* d_in;  size_t in_cap;
    void* d_out; size_t out_cap;
    char* h_in;  size_t hin_cap;        /* page-locked staging: the copies to and from them are truly asynchronous */
    char* h_out; size_t hout_cap;
    int* rows;                          /* page-locked, read / written by the kernels: 3 plans x {source size, capacity, result} */
    lz4amd_plan* plan[3];               /* by lz4amd_op: COMPRESS, DECOMPRESS, COMPRESS_HC - bound to d_in / d_out */
} lz4amd_thread_slot;
static pthread_key_t g_slot_key;
static pthread_once_t g_slot_once = PTHREAD_ONCE_INIT;
static void slot_free(void* v)
{
    lz4amd_thread_slot* t = (lz4amd_thread_slot*)v;
    int i;
    if (!t) return;
    for (i = 0; i < 3; i++) lz4amd_plan_destroy(t->plan[i]);
    lz4amd_hip_free(t->d_in); lz4amd_hip_free(t->d_out);
    lz4amd_hip_host_free(t->h_in); lz4amd_hip_host_free(t->h_out); lz4amd_hip_host_free(t->rows);
    lz4amd_hip_stream_destroy(t->stream);
    free(t);
}
static void slot_key_init(void) { (void)pthread_key_create(&g_slot_key, slot_free); }
static lz4amd_thread_slot* slot_get(void)
{
    lz4amd_thread_slot* t;
    pthread_once(&g_slot_once, slot_key_init);
    t = (lz4amd_thread_slot*)pthread_getspecific(g_slot_key);
    if (!t) {
        t = (lz4amd_thread_slot*)calloc(1, sizeof *t);
        if (!t) return NULL;
        t->stream = lz4amd_hip_stream_create();
        t->rows = (int*)lz4amd_hip_host_alloc(16 * sizeof(int));
        if (!t->stream || !t->rows || pthread_setspecific(g_slot_key, t)) { slot_free(t); return NULL; }
    }
    return t;
}

/* History (LZ4_decompress_safe_usingDict, the streaming contexts): up to 64 KB that sit in front of the block in device
 * memory - before the SOURCE for the compressors (lz4.c:1707), before the OUTPUT for the decoder (lz4.c:2719-2732: prefix
 * mode; a dictionary elsewhere in host memory needs no separate code path on the device).  The thread's buffers keep 64 KB of
 * room in front of the block for it and its plans carry a history column, so a call with history costs what one without does:
 * no lock, no device allocation, no plan construction. */
#define HIST_ROOM 65536u
static int run_one_hist(lz4amd_op op, const char* hist, int histSize, const char* src, char* dst, int srcSize, int dstCapacity, int level, int fail)
{
    lz4amd_ctx* ctx;
    lz4amd_thread_slot* t;
    int result = fail, i;
    size_t in_bytes = srcSize > 0 ? (size_t)srcSize : 0;
    size_t out_bytes = dstCapacity > 0 ? (size_t)dstCapacity : 0;
    const size_t pre = (hist && histSize > 0) ? ((size_t)histSize > HIST_ROOM ? HIST_ROOM : (size_t)histSize) : 0;
    if ((int)op < 0 || (int)op > 2) return fail;

    pthread_mutex_lock(&g_lock);                     /* (only the first call creates the context) */
    ctx = lz4amd_default_ctx();
    pthread_mutex_unlock(&g_lock);
    if (!ctx || lz4amd_hip_use_device(ctx->device)) return fail;
    t = slot_get();
    if (!t) return fail;
    if (HIST_ROOM + in_bytes + 16 > t->in_cap || HIST_ROOM + out_bytes + 16 > t->out_cap || HIST_ROOM + in_bytes + 16 > t->hin_cap || HIST_ROOM + out_bytes + 16 > t->hout_cap) {
        /* the buffers grow: the plans bound to them go (rare: sizes settle after the first blocks).
         * (the page-locked buffers are part of the condition: a failed allocation must not leave a NULL staging pointer behind
         *  device buffers that look large enough to the next call) */
        for (i = 0; i < 3; i++) { lz4amd_plan_destroy(t->plan[i]); t->plan[i] = NULL; }
        if (stage_reserve(&t->d_in, &t->in_cap, HIST_ROOM + in_bytes + 16) || stage_reserve(&t->d_out, &t->out_cap, HIST_ROOM + out_bytes + 16)) return fail;
        if (t->in_cap > t->hin_cap) { lz4amd_hip_host_free(t->h_in); t->h_in = (char*)lz4amd_hip_host_alloc(t->in_cap); t->hin_cap = t->h_in ? t->in_cap : 0; }
        if (t->out_cap > t->hout_cap) { lz4amd_hip_host_free(t->h_out); t->h_out = (char*)lz4amd_hip_host_alloc(t->out_cap); t->hout_cap = t->h_out ? t->out_cap : 0; }
        if (!t->h_in || !t->h_out) return fail;
    }
    if (!t->plan[op]) {
        /* sized for the largest block the buffers take (decoder / HC scratch follow the source size) */
        const void* dsrc = (char*)t->d_in + HIST_ROOM; void* ddst = (char*)t->d_out + HIST_ROOM;
        const size_t smax_z = t->in_cap - HIST_ROOM - 16, cmax_z = t->out_cap - HIST_ROOM - 16;
        int smax = smax_z > 0x7E000000u ? 0x7E000000 : (int)smax_z, cmax = cmax_z > 0x7FFFFFFFu ? 0x7FFFFFFF : (int)cmax_z, pmax = (int)HIST_ROOM;
        int rc = op == LZ4AMD_OP_DECOMPRESS ? lz4amd_plan_create_prefix(ctx, &t->plan[op], 1, &dsrc, &smax, &ddst, &cmax, &pmax)
               : op == LZ4AMD_OP_COMPRESS   ? lz4amd_plan_create_compress_prefix(ctx, &t->plan[op], 1, &dsrc, &smax, &ddst, &cmax, &pmax)
                                            : lz4amd_plan_create_compress_hc_prefix(ctx, &t->plan[op], 1, &dsrc, &smax, &ddst, &cmax, &pmax, level);
        if (rc) return fail;
        if (lz4amd_plan_bind_host_row(t->plan[op], t->rows + 4 * (int)op)) return fail;
    }
    {
        int* row = t->rows + 4 * (int)op;
        row[0] = srcSize; row[1] = dstCapacity; row[2] = fail; row[3] = (int)pre;
        lz4amd_plan_set_level(t->plan[op], level);
        if (op == LZ4AMD_OP_DECOMPRESS) {
            if (pre) { memcpy(t->h_out + HIST_ROOM - pre, hist + ((size_t)histSize - pre), pre); if (lz4amd_hip_h2d((char*)t->d_out + HIST_ROOM - pre, t->h_out + HIST_ROOM - pre, pre, t->stream)) return fail; }
            if (in_bytes) { memcpy(t->h_in + HIST_ROOM, src, in_bytes); if (lz4amd_hip_h2d((char*)t->d_in + HIST_ROOM, t->h_in + HIST_ROOM, in_bytes, t->stream)) return fail; }
        } else {
            if (pre) memcpy(t->h_in + HIST_ROOM - pre, hist + ((size_t)histSize - pre), pre);
            if (in_bytes) memcpy(t->h_in + HIST_ROOM, src, in_bytes);
            if (pre + in_bytes && lz4amd_hip_h2d((char*)t->d_in + HIST_ROOM - pre, t->h_in + HIST_ROOM - pre, pre + in_bytes, t->stream)) return fail;
        }
        if (lz4amd_plan_launch(t->plan[op], t->stream) || lz4amd_hip_sync(t->stream)) return fail;
        result = row[2];
        if (result > 0 && (size_t)result <= out_bytes) {
            if (lz4amd_hip_d2h(t->h_out + HIST_ROOM, (char*)t->d_out + HIST_ROOM, (size_t)result, t->stream) || lz4amd_hip_sync(t->stream)) return fail;
            memcpy(dst, t->h_out + HIST_ROOM, (size_t)result);
        }
    }
    return result;
}
int lz4amd_run_one(lz4amd_op op, const char* src, char* dst, int srcSize, int dstCapacity, int level, int fail)
{
    return run_one_hist(op, NULL, 0, src, dst, srcSize, dstCapacity, level, fail);
}

/* lz4.c:1453 LZ4_compress_fast: `acceleration` trades ratio for speed.  In the reference's serial probe loop it is the
 * initial step between probed positions (lz4.c:1044-1053, clamped lz4.c:1386-1387).  Here every probe of a tile runs at
 * once, so the knob has two settings: 1 probes every second position of a block of 64 KB or more (every position of a
 * smaller one), 2 and above every fourth.  Sizes never shrink as the value grows; at 1 they are within 3 % of the
 * reference's, at 2 within 5 % of the reference's at 2 (tests/test_gpu_parity.py).  What it buys depends on the data: fewer
 * sequences to select and emit make highly compressible input 20 % faster, datagen -P60 no faster. */
int LZ4_compress_fast(const char* src, char* dst, int srcSize, int dstCapacity, int acceleration)
{
    lz4amd_set_notice(acceleration > 2 ? "LZ4_compress_fast: acceleration > 2 is parsed as acceleration 2 (every fourth position is probed)" : "");
    if (srcSize < 0 || (unsigned)srcSize > (unsigned)LZ4_MAX_INPUT_SIZE) return 0;   /* lz4.c:1360 */
    if (dst == NULL || dstCapacity <= 0) return 0;
    if (src == NULL && srcSize != 0) return 0;
    return lz4amd_run_one(LZ4AMD_OP_COMPRESS, src, dst, srcSize, dstCapacity, acceleration, 0);
}

int LZ4_compress_default(const char* src, char* dst, int srcSize, int dstCapacity)
{   /* lz4.c:1472 */
    return LZ4_compress_fast(src, dst, srcSize, dstCapacity, 1);
}

/* lz4.c:1382: the caller-provided state is not needed by the device path (tables live in LDS);
 * it is accepted for ABI compatibility. */
int LZ4_sizeofState(void) { return LZ4_STREAM_MINSIZE; }
int LZ4_compress_fast_extState(void* state, const char* src, char* dst, int srcSize, int dstCapacity, int acceleration)
{
    if (state == NULL) return 0;
    return LZ4_compress_fast(src, dst, srcSize, dstCapacity, acceleration);
}

int LZ4_decompress_safe(const char* src, char* dst, int compressedSize, int dstCapacity)
{   /* lz4.c:2451; degenerate cases lz4.c:2036, 2062-2069 decided by the kernel itself */
    if (src == NULL || dstCapacity < 0) return -1;
    if (compressedSize < 0) return -1;
    return lz4amd_run_one(LZ4AMD_OP_DECOMPRESS, src, dst, compressedSize, dstCapacity, 0, -1);
}

/* lz4.c:2719-2732 LZ4_decompress_safe_usingDict: the dictionary (its last 64 KB) is staged right
 * before the output in device memory, which is the decoder's prefix mode (lz4.c:2479 / 2504); a
 * dictionary elsewhere in memory (the reference's extDict branches, lz4.c:2166-2196) needs no
 * separate code path on the device. */
int LZ4_decompress_safe_usingDict(const char* src, char* dst, int compressedSize, int dstCapacity,
                                  const char* dictStart, int dictSize)
{
    if (src == NULL || dstCapacity < 0 || compressedSize < 0) return -1;
    if (dictStart == NULL || dictSize <= 0) return LZ4_decompress_safe(src, dst, compressedSize, dstCapacity);
    return run_one_hist(LZ4AMD_OP_DECOMPRESS, dictStart, dictSize, src, dst, compressedSize, dstCapacity, 0, -1);
}

/* One block with up to 64 KB of history (the bytes a streaming compressor may reference, lz4.c:1707
 * LZ4_compress_fast_continue): history and block are staged back to back in device memory and the block
 * is compressed with the history as its prefix (include/lz4amd.h lz4amd_plan_create_compress_prefix). */
int lz4amd_compress_with_history(const char* hist, int histSize, const char* src, char* dst, int srcSize, int dstCapacity, int hc_level)
{   /* hc_level 0: LZ4_compress_default semantics; > 0: LZ4_compress_HC at that level; < 0: LZ4_compress_fast with acceleration -hc_level */
    if (srcSize < 0 || (unsigned)srcSize > (unsigned)LZ4_MAX_INPUT_SIZE || dst == NULL || dstCapacity <= 0) return 0;
    if (src == NULL && srcSize != 0) return 0;
    return run_one_hist(hc_level > 0 ? LZ4AMD_OP_COMPRESS_HC : LZ4AMD_OP_COMPRESS, hist, histSize, src, dst, srcSize, dstCapacity, hc_level > 0 ? hc_level : -hc_level, 0);
}
