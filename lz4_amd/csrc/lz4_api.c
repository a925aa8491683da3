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

pthread_mutex_t lz4amd_default_lock = PTHREAD_MUTEX_INITIALIZER;   /* shared with lz4frame_api.c */
#define g_lock lz4amd_default_lock
static lz4amd_ctx* g_ctx = NULL;
static int g_ctx_failed = 0;
/* grow-only device staging buffers of the default context (guarded by g_lock) */
static void* g_stage_in = NULL;  static size_t g_stage_in_cap = 0;
static void* g_stage_out = NULL; static size_t g_stage_out_cap = 0;

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

/* one block through the device; op selects the kernel set.  Returns the per-block result, or
 * `fail` when the device path cannot run. */
int lz4amd_run_one(lz4amd_op op, const char* src, char* dst, int srcSize, int dstCapacity, int level, int fail)
{
    lz4amd_ctx* ctx;
    lz4amd_plan* plan = NULL;
    int result = fail, rc;
    const void* dsrc; void* ddst;
    size_t in_bytes = srcSize > 0 ? (size_t)srcSize : 0;
    size_t out_bytes = dstCapacity > 0 ? (size_t)dstCapacity : 0;

    pthread_mutex_lock(&g_lock);
    ctx = lz4amd_default_ctx();
    if (!ctx) goto done;
    if (stage_reserve(&g_stage_in, &g_stage_in_cap, in_bytes + 16) ||
        stage_reserve(&g_stage_out, &g_stage_out_cap, out_bytes + 16)) goto done;
    if (in_bytes && lz4amd_hip_h2d(g_stage_in, src, in_bytes, NULL)) goto done;
    dsrc = g_stage_in; ddst = g_stage_out;
    rc = lz4amd_plan_create(ctx, &plan, op, 1, &dsrc, &srcSize, &ddst, &dstCapacity, level);
    if (rc) goto done;
    if (lz4amd_plan_launch(plan, NULL) || lz4amd_plan_results(plan, &result, NULL)) { result = fail; goto done; }
    if (result > 0 && (size_t)result <= out_bytes) {
        if (lz4amd_hip_d2h(dst, g_stage_out, (size_t)result, NULL) || lz4amd_hip_sync(NULL)) result = fail;
    }
done:
    lz4amd_plan_destroy(plan);
    pthread_mutex_unlock(&g_lock);
    return result;
}

/* lz4.c:1453 LZ4_compress_fast: `acceleration` trades ratio for speed in the reference's serial
 * probe loop (lz4.c:1044-1053); the wave-parallel matcher probes every position at no extra cost,
 * so the value is accepted and ignored (any value yields a valid block). */
int LZ4_compress_fast(const char* src, char* dst, int srcSize, int dstCapacity, int acceleration)
{
    (void)acceleration;
    if (srcSize < 0 || (unsigned)srcSize > (unsigned)LZ4_MAX_INPUT_SIZE) return 0;   /* lz4.c:1360 */
    if (dst == NULL || dstCapacity <= 0) return 0;
    if (src == NULL && srcSize != 0) return 0;
    return lz4amd_run_one(LZ4AMD_OP_COMPRESS, src, dst, srcSize, dstCapacity, 0, 0);
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
    lz4amd_ctx* ctx;
    lz4amd_plan* plan = NULL;
    int result = -1, pre;
    const void* dsrc; void* ddst;
    size_t in_bytes, out_bytes;
    if (src == NULL || dstCapacity < 0 || compressedSize < 0) return -1;
    if (dictStart == NULL || dictSize <= 0) return LZ4_decompress_safe(src, dst, compressedSize, dstCapacity);
    pre = dictSize > 65536 ? 65536 : dictSize;
    in_bytes = (size_t)compressedSize; out_bytes = (size_t)dstCapacity;
    pthread_mutex_lock(&g_lock);
    ctx = lz4amd_default_ctx();
    if (!ctx) goto done;
    if (stage_reserve(&g_stage_in, &g_stage_in_cap, in_bytes + 16) ||
        stage_reserve(&g_stage_out, &g_stage_out_cap, out_bytes + 65536 + 16)) goto done;
    if (in_bytes && lz4amd_hip_h2d(g_stage_in, src, in_bytes, NULL)) goto done;
    if (lz4amd_hip_h2d((char*)g_stage_out + (65536 - pre), dictStart + (dictSize - pre), (size_t)pre, NULL)) goto done;
    dsrc = g_stage_in; ddst = (char*)g_stage_out + 65536;
    if (lz4amd_plan_create_prefix(ctx, &plan, 1, &dsrc, &compressedSize, &ddst, &dstCapacity, &pre)) goto done;
    if (lz4amd_plan_launch(plan, NULL) || lz4amd_plan_results(plan, &result, NULL)) { result = -1; goto done; }
    if (result > 0 && (size_t)result <= out_bytes) {
        if (lz4amd_hip_d2h(dst, ddst, (size_t)result, NULL) || lz4amd_hip_sync(NULL)) result = -1;
    }
done:
    lz4amd_plan_destroy(plan);
    pthread_mutex_unlock(&g_lock);
    return result;
}

/* One block with up to 64 KB of history (the bytes a streaming compressor may reference, lz4.c:1707
 * LZ4_compress_fast_continue): history and block are staged back to back in device memory and the block
 * is compressed with the history as its prefix (include/lz4amd.h lz4amd_plan_create_compress_prefix). */
int lz4amd_compress_with_history(const char* hist, int histSize, const char* src, char* dst, int srcSize, int dstCapacity, int hc_level)
{   /* hc_level 0: LZ4_compress_default semantics; > 0: LZ4_compress_HC at that level */
    lz4amd_ctx* ctx;
    lz4amd_plan* plan = NULL;
    int result = 0, pre;
    const void* dsrc; void* ddst;
    size_t in_bytes, out_bytes;
    if (srcSize < 0 || (unsigned)srcSize > (unsigned)LZ4_MAX_INPUT_SIZE || dst == NULL || dstCapacity <= 0) return 0;
    if (src == NULL && srcSize != 0) return 0;
    if (hist == NULL || histSize <= 0)
        return lz4amd_run_one(hc_level > 0 ? LZ4AMD_OP_COMPRESS_HC : LZ4AMD_OP_COMPRESS, src, dst, srcSize, dstCapacity, hc_level, 0);
    pre = histSize > 65536 ? 65536 : histSize;
    in_bytes = (size_t)srcSize; out_bytes = (size_t)dstCapacity;
    pthread_mutex_lock(&g_lock);
    ctx = lz4amd_default_ctx();
    if (!ctx) goto done;
    if (stage_reserve(&g_stage_in, &g_stage_in_cap, in_bytes + 65536 + 16) ||
        stage_reserve(&g_stage_out, &g_stage_out_cap, out_bytes + 16)) goto done;
    if (lz4amd_hip_h2d((char*)g_stage_in + (65536 - pre), hist + (histSize - pre), (size_t)pre, NULL)) goto done;
    if (in_bytes && lz4amd_hip_h2d((char*)g_stage_in + 65536, src, in_bytes, NULL)) goto done;
    dsrc = (char*)g_stage_in + 65536; ddst = g_stage_out;
    if (hc_level > 0 ? lz4amd_plan_create_compress_hc_prefix(ctx, &plan, 1, &dsrc, &srcSize, &ddst, &dstCapacity, &pre, hc_level)
                     : lz4amd_plan_create_compress_prefix(ctx, &plan, 1, &dsrc, &srcSize, &ddst, &dstCapacity, &pre)) goto done;
    if (lz4amd_plan_launch(plan, NULL) || lz4amd_plan_results(plan, &result, NULL)) { result = 0; goto done; }
    if (result > 0 && (size_t)result <= out_bytes) {
        if (lz4amd_hip_d2h(dst, g_stage_out, (size_t)result, NULL) || lz4amd_hip_sync(NULL)) result = 0;
    }
done:
    lz4amd_plan_destroy(plan);
    pthread_mutex_unlock(&g_lock);
    return result;
}
