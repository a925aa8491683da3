/*
 * lz4_compat_api.c -- the long tail of the reference's block ABI (lib/lz4.h, lib/lz4hc.h): the names the reference's
 * own test programs (tests/fuzzer.c, tests/frametest.c) and its CLI link against besides the hot path.
 *
 * Three kinds of entry points live here:
 *   (1) GPU-backed: names that are another spelling of a call the device path already serves (_fastReset,
 *       attach_dictionary, withPrefix64k, forceExtDict, the deprecated compress wrappers), and the _destSize family,
 *       which finds the longest prefix of the input that fits the destination by compressing prefixes on the device
 *       (bisection: compressed size grows with the input, give or take a tile).
 *   (2) host C written from the block format document (doc/lz4_Block_format.md): LZ4_decompress_safe_partial* and the
 *       deprecated LZ4_decompress_fast* family.  Neither has a device kernel: "partial" stops in the middle of a block
 *       whose full size the caller does not provide, "fast" does not know where its input ends.  They are NOT a
 *       fallback of the hot path - LZ4_decompress_safe / LZ4_compress_* never come here - and they share no code with
 *       oracle/ (test infrastructure).
 *   (3) xxHash under the names the reference's shared library exports (lib/Makefile:49 XXH_NAMESPACE=LZ4_), from the
 *       xxHash specification.
 */
#include "../../include/lz4.h"
#include "../../include/lz4hc.h"
#include "lz4amd_internal.h"
#include "xxh32_host.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ (1) other spellings of device-backed calls */
int LZ4_compress_fast_extState_fastReset(void* state, const char* src, char* dst, int srcSize, int dstCapacity, int acceleration)
{   /* lz4.c:1420: "state" only has to be valid; there is no table to reset on the host */
    return LZ4_compress_fast_extState(state, src, dst, srcSize, dstCapacity, acceleration);
}
int LZ4_compress_HC_extStateHC_fastReset(void* state, const char* src, char* dst, int srcSize, int dstCapacity, int compressionLevel)
{   /* lz4hc.c:1492 */
    return LZ4_compress_HC_extStateHC(state, src, dst, srcSize, dstCapacity, compressionLevel);
}
void LZ4_attach_dictionary(LZ4_stream_t* workingStream, const LZ4_stream_t* dictionaryStream)
{   /* lz4.c:1658-1684: the working stream compresses its next block against the dictionary stream's content without
     * copying tables.  A context here is only the location of the history, so attaching is pointing at it. */
    if (workingStream == NULL) return;
    if (dictionaryStream != NULL && dictionaryStream->internal_donotuse.dictSize) {
        workingStream->internal_donotuse.dictionary = dictionaryStream->internal_donotuse.dictionary;
        workingStream->internal_donotuse.dictSize = dictionaryStream->internal_donotuse.dictSize;
    } else { workingStream->internal_donotuse.dictionary = NULL; workingStream->internal_donotuse.dictSize = 0; }
}
void LZ4_attach_HC_dictionary(LZ4_streamHC_t* workingStream, const LZ4_streamHC_t* dictionaryStream)
{   /* lz4hc.c:1642-1646 */
    if (workingStream == NULL) return;
    if (dictionaryStream != NULL && dictionaryStream->internal_donotuse.dictSize) {
        workingStream->internal_donotuse.dictionary = dictionaryStream->internal_donotuse.dictionary;
        workingStream->internal_donotuse.dictSize = dictionaryStream->internal_donotuse.dictSize;
    } else { workingStream->internal_donotuse.dictionary = NULL; workingStream->internal_donotuse.dictSize = 0; }
}
int LZ4_loadDictSlow(LZ4_stream_t* s, const char* dictionary, int dictSize) { return LZ4_loadDict(s, dictionary, dictSize); }   /* lz4.c:1621: the table is built on the device either way */
void LZ4_favorDecompressionSpeed(LZ4_streamHC_t* s, int favor)
{   /* lz4hc.c:1621: a preference of the optimal parser (levels 10-12); it travels to the kernel with the level (LZ4AMD_HC_FAVOR_DEC_SPEED) */
    if (s) s->internal_donotuse.favorDecSpeed = (signed char)(favor != 0);
}
int LZ4_decompress_safe_withPrefix64k(const char* src, char* dst, int compressedSize, int maxOutputSize)
{   /* lz4.c:2479: 64 KB of history sit right before dst */
    return LZ4_decompress_safe_usingDict(src, dst, compressedSize, maxOutputSize, dst - 65536, 65536);
}
int LZ4_decompress_safe_forceExtDict(const char* src, char* dst, int compressedSize, int maxOutputSize, const void* dictStart, size_t dictSize)
{   /* lz4.c:2545 (a test hook of the reference: the dictionary is treated as external even when it is contiguous) */
    return LZ4_decompress_safe_usingDict(src, dst, compressedSize, maxOutputSize, (const char*)dictStart, dictSize > 0x7FFFFFFF ? 0x7FFFFFFF : (int)dictSize);
}
int LZ4_compress_forceExtDict(LZ4_stream_t* s, const char* src, char* dst, int srcSize)
{   /* lz4.c:1784 (test hook): no capacity limit */
    return LZ4_compress_fast_continue(s, src, dst, srcSize, LZ4_compressBound(srcSize), 1);
}

/* ---- _destSize (lz4.c:1506-1541, lz4hc.c:1527-1536, 1718-1721): as much of the input as fits `target` bytes */
typedef int (*prefix_fn)(void* arg, const char* src, char* dst, int n, int cap);
static int dest_size_search(prefix_fn f, void* arg, const char* src, char* dst, int* srcSizePtr, int target)
{
    int lo, hi, best = 0, r;
    const int srcSize = *srcSizePtr;
    if (target <= 0 || srcSize < 0) { *srcSizePtr = 0; return 0; }
    r = f(arg, src, dst, srcSize, target);                     /* everything fits? (the usual case when target >= the bound) */
    if (r > 0) return r;
    lo = 0; hi = srcSize;                                      /* f(lo) fits (an empty block is one byte), f(hi) does not */
    while (hi - lo > 1) {
        const int mid = lo + (hi - lo) / 2;
        r = f(arg, src, dst, mid, target);
        if (r > 0) { lo = mid; best = r; } else hi = mid;
    }
    r = f(arg, src, dst, lo, target);                          /* (the last probe may have been a failing one: dst holds garbage) */
    if (r <= 0) { *srcSizePtr = 0; return 0; }
    (void)best;
    *srcSizePtr = lo;
    return r;
}
static int fast_prefix(void* arg, const char* src, char* dst, int n, int cap) { return LZ4_compress_fast(src, dst, n, cap, *(const int*)arg); }
int LZ4_compress_destSize(const char* src, char* dst, int* srcSizePtr, int targetDstSize)
{
    int accel = 1;
    if (srcSizePtr == NULL) return 0;
    return dest_size_search(fast_prefix, &accel, src, dst, srcSizePtr, targetDstSize);
}
int LZ4_compress_destSize_extState(void* state, const char* src, char* dst, int* srcSizePtr, int targetDstSize, int acceleration)
{   /* lz4.c:1506: the same parse as LZ4_compress_fast_extState at this acceleration (fuzzer.c:488-492: one byte less room than that
     * call needed must take less than the whole input) */
    if (state == NULL || srcSizePtr == NULL) return 0;
    return dest_size_search(fast_prefix, &acceleration, src, dst, srcSizePtr, targetDstSize);
}
static int hc_prefix(void* arg, const char* src, char* dst, int n, int cap) { return LZ4_compress_HC(src, dst, n, cap, *(const int*)arg); }
int LZ4_compress_HC_destSize(void* stateHC, const char* src, char* dst, int* srcSizePtr, int targetDstSize, int compressionLevel)
{   /* lz4hc.c:1527 */
    if (stateHC == NULL || srcSizePtr == NULL) return 0;
    return dest_size_search(hc_prefix, &compressionLevel, src, dst, srcSizePtr, targetDstSize);
}
typedef struct { LZ4_streamHC_t* s; const char* dictionary; unsigned dictSize; int level; } hc_cont_arg;
static int hc_cont_prefix(void* arg, const char* src, char* dst, int n, int cap)
{   /* every probe starts from the stream's state before the call */
    hc_cont_arg* a = (hc_cont_arg*)arg;
    a->s->internal_donotuse.dictionary = a->dictionary; a->s->internal_donotuse.dictSize = a->dictSize; a->s->internal_donotuse.compressionLevel = a->level;
    return LZ4_compress_HC_continue(a->s, src, dst, n, cap);
}
int LZ4_compress_HC_continue_destSize(LZ4_streamHC_t* s, const char* src, char* dst, int* srcSizePtr, int targetDstSize)
{   /* lz4hc.c:1718 */
    hc_cont_arg a;
    if (s == NULL || srcSizePtr == NULL) return 0;
    a.s = s; a.dictionary = s->internal_donotuse.dictionary; a.dictSize = s->internal_donotuse.dictSize; a.level = s->internal_donotuse.compressionLevel;
    return dest_size_search(hc_cont_prefix, &a, src, dst, srcSizePtr, targetDstSize);
}

/* ------------------------------------------------------------------ (2) host decoders without a device counterpart
 * One sequential decoder from the block format document.  out(i), i < 0, is the history: the `dictSize` bytes that end
 * at dictEnd (for a prefix, dictEnd == dst).  Modes: partial (stop when `cap` bytes are out or the input is used up),
 * fast (the input's end is unknown, the output must come out exactly `cap` bytes long). */
static int host_decode(const uint8_t* src, int srcSize, uint8_t* dst, int cap, int partial, int fast, const uint8_t* dictEnd, size_t dictSize)
{
    const uint8_t* ip = src;
    const uint8_t* const iend = src + (fast ? 0 : srcSize);          /* (not used in fast mode) */
    uint8_t* op = dst;
    uint8_t* const oend = dst + cap;
    if (src == NULL || cap < 0 || (!fast && srcSize <= 0)) return -1;
    if (cap == 0) { if (fast) return src[0] == 0 ? 1 : -1; return partial ? 0 : -1; }
    for (;;) {
        unsigned token;
        size_t ll, ml, offset, room;
        if (!fast && ip >= iend) { if (partial) break; return -1; }
        token = *ip++;
        ll = token >> 4;
        if (ll == 15) {
            unsigned b;
            do { if (!fast && ip >= iend) return -1; b = *ip++; ll += b; if (ll > ((size_t)1 << 31)) return -1; } while (b == 255);
        }
        room = (size_t)(oend - op);
        if (fast) {
            /* lz4.c:2276-2330: the run fits; one that ends within the last 12 output bytes is the block's last and ends exactly there */
            if (ll > room || (room - ll < 12 && ll != room)) return -1;
            memmove(op, ip, ll); op += ll; ip += ll;
            if (op == oend) break;
        } else {
            size_t n = ll;
            if (n > room) n = room;
            if (n > (size_t)(iend - ip)) n = (size_t)(iend - ip);
            memmove(op, ip, n); op += n; ip += n;
            if (n < ll || ip >= iend || op == oend) break;                /* output full, input used up, or the block's last literals */
            if (iend - ip <= 2) break;                           /* (lz4.c:2325: a truncated input ends before an offset with nothing behind it) */
        }
        offset = (size_t)ip[0] | ((size_t)ip[1] << 8); ip += 2;
        ml = token & 15;
        if (ml == 15) {
            unsigned b;
            do { if (!fast && ip >= iend) return -1; b = *ip++; ml += b; if (ml > ((size_t)1 << 31)) return -1; } while (b == 255);
        }
        ml += 4;
        if (offset == 0 || offset > (size_t)(op - dst) + dictSize) return -1;        /* lz4.c:2356 */
        room = (size_t)(oend - op);
        if (fast) { if (room < 5 || ml > room - 5) return -1; }                        /* the last 5 bytes are literals (lz4.c:2423) */
        else if (ml > room) ml = room;
        {
            size_t i;
            for (i = 0; i < ml; i++, op++) {
                const ptrdiff_t pos = (op - dst) - (ptrdiff_t)offset;
                *op = pos >= 0 ? dst[pos] : dictEnd[pos];
            }
        }
        if (!fast && op == oend) break;
    }
    if (fast) return (int)(ip - src);
    return (int)(op - dst);
}

int LZ4_decompress_safe_partial(const char* src, char* dst, int compressedSize, int targetOutputSize, int dstCapacity)
{   /* lz4.c:2459 */
    int cap = targetOutputSize < dstCapacity ? targetOutputSize : dstCapacity;
    if (cap < 0) return -1;
    return host_decode((const uint8_t*)src, compressedSize, (uint8_t*)dst, cap, 1, 0, (const uint8_t*)dst, 0);
}
int LZ4_decompress_safe_partial_usingDict(const char* src, char* dst, int compressedSize, int targetOutputSize, int dstCapacity,
                                          const char* dictStart, int dictSize)
{   /* lz4.c:2734 */
    int cap = targetOutputSize < dstCapacity ? targetOutputSize : dstCapacity;
    if (cap < 0) return -1;
    if (dictStart == NULL || dictSize <= 0) return host_decode((const uint8_t*)src, compressedSize, (uint8_t*)dst, cap, 1, 0, (const uint8_t*)dst, 0);
    return host_decode((const uint8_t*)src, compressedSize, (uint8_t*)dst, cap, 1, 0, (const uint8_t*)dictStart + dictSize, (size_t)dictSize);
}
int LZ4_decompress_safe_partial_forceExtDict(const char* src, char* dst, int compressedSize, int targetOutputSize, int dstCapacity,
                                             const void* dictStart, size_t dictSize)
{   /* lz4.c:2555 (test hook) */
    return LZ4_decompress_safe_partial_usingDict(src, dst, compressedSize, targetOutputSize, dstCapacity, (const char*)dictStart, dictSize > 0x7FFFFFFF ? 0x7FFFFFFF : (int)dictSize);
}
/* deprecated (lz4.h:806-826): the caller vouches for the input; returns the number of input bytes read */
int LZ4_decompress_fast(const char* src, char* dst, int originalSize)
{   /* lz4.c:2470 */
    return host_decode((const uint8_t*)src, 0, (uint8_t*)dst, originalSize, 0, 1, (const uint8_t*)dst, 0);
}
int LZ4_decompress_fast_withPrefix64k(const char* src, char* dst, int originalSize)
{   /* lz4.c:2488 */
    return host_decode((const uint8_t*)src, 0, (uint8_t*)dst, originalSize, 0, 1, (const uint8_t*)dst, 65536);
}
int LZ4_decompress_fast_usingDict(const char* src, char* dst, int originalSize, const char* dictStart, int dictSize)
{   /* lz4.c:2749 */
    if (dictStart == NULL || dictSize <= 0) return LZ4_decompress_fast(src, dst, originalSize);
    return host_decode((const uint8_t*)src, 0, (uint8_t*)dst, originalSize, 0, 1, (const uint8_t*)dictStart + dictSize, (size_t)dictSize);
}
int LZ4_decompress_fast_continue(LZ4_streamDecode_t* sd, const char* src, char* dst, int originalSize)
{   /* lz4.c:2681-2716: prefix / external dictionary bookkeeping as in LZ4_decompress_safe_continue */
    int r;
    if (sd == NULL || originalSize < 0) return -1;
    if (sd->internal_donotuse.prefixSize == 0) {
        r = LZ4_decompress_fast(src, dst, originalSize);
        if (r <= 0) return r;
        sd->internal_donotuse.prefixSize = (size_t)originalSize;
        sd->internal_donotuse.prefixEnd = (const unsigned char*)dst + originalSize;
    } else if (sd->internal_donotuse.prefixEnd == (const unsigned char*)dst) {
        /* rolling prefix: what is reachable is the prefix, then the older external segment */
        const size_t ps = sd->internal_donotuse.prefixSize;
        if (ps >= 65535 || sd->internal_donotuse.extDictSize == 0)
            r = host_decode((const uint8_t*)src, 0, (uint8_t*)dst, originalSize, 0, 1, (const uint8_t*)dst, ps);
        else {
            /* short prefix + external segment: gather both into one history buffer */
            const size_t es = sd->internal_donotuse.extDictSize;
            const size_t take = es > 65536 - ps ? 65536 - ps : es;
            uint8_t* hist = (uint8_t*)malloc(take + ps);
            if (!hist) return -1;
            memcpy(hist, sd->internal_donotuse.externalDict + (es - take), take);
            memcpy(hist + take, (const uint8_t*)dst - ps, ps);
            r = host_decode((const uint8_t*)src, 0, (uint8_t*)dst, originalSize, 0, 1, hist + take + ps, take + ps);
            free(hist);
        }
        if (r <= 0) return r;
        sd->internal_donotuse.prefixSize += (size_t)originalSize;
        sd->internal_donotuse.prefixEnd += originalSize;
    } else {
        sd->internal_donotuse.extDictSize = sd->internal_donotuse.prefixSize;
        sd->internal_donotuse.externalDict = sd->internal_donotuse.prefixEnd - sd->internal_donotuse.extDictSize;
        {   /* the LAST 64 KB of what was decoded before (lz4.c:2817-2825: the dictionary ends where the previous prefix ended) */
            const size_t es = sd->internal_donotuse.extDictSize, use = es > 65536 ? 65536 : es;
            r = LZ4_decompress_fast_usingDict(src, dst, originalSize, (const char*)sd->internal_donotuse.externalDict + (es - use), (int)use);
        }
        if (r <= 0) return r;
        sd->internal_donotuse.prefixSize = (size_t)originalSize;
        sd->internal_donotuse.prefixEnd = (const unsigned char*)dst + originalSize;
    }
    return r;
}

/* ---- deprecated names (lz4.h:784-845, lz4hc.h:290-352): thin wrappers */
int LZ4_compress(const char* src, char* dst, int srcSize) { return LZ4_compress_default(src, dst, srcSize, LZ4_compressBound(srcSize)); }
int LZ4_compress_limitedOutput(const char* src, char* dst, int srcSize, int maxOutputSize) { return LZ4_compress_default(src, dst, srcSize, maxOutputSize); }
int LZ4_compress_withState(void* state, const char* src, char* dst, int srcSize) { return LZ4_compress_fast_extState(state, src, dst, srcSize, LZ4_compressBound(srcSize), 1); }
int LZ4_compress_limitedOutput_withState(void* state, const char* src, char* dst, int srcSize, int maxOutputSize) { return LZ4_compress_fast_extState(state, src, dst, srcSize, maxOutputSize, 1); }
int LZ4_compress_continue(LZ4_stream_t* s, const char* src, char* dst, int srcSize) { return LZ4_compress_fast_continue(s, src, dst, srcSize, LZ4_compressBound(srcSize), 1); }
int LZ4_compress_limitedOutput_continue(LZ4_stream_t* s, const char* src, char* dst, int srcSize, int maxOutputSize) { return LZ4_compress_fast_continue(s, src, dst, srcSize, maxOutputSize, 1); }
int LZ4_uncompress(const char* src, char* dst, int outputSize) { return LZ4_decompress_fast(src, dst, outputSize); }
int LZ4_uncompress_unknownOutputSize(const char* src, char* dst, int isize, int maxOutputSize) { return LZ4_decompress_safe(src, dst, isize, maxOutputSize); }
int LZ4_sizeofStreamState(void) { return (int)sizeof(LZ4_stream_t); }
int LZ4_resetStreamState(void* state, char* inputBuffer) { (void)inputBuffer; LZ4_resetStream((LZ4_stream_t*)state); return 0; }
void* LZ4_create(char* inputBuffer) { (void)inputBuffer; return LZ4_createStream(); }
char* LZ4_slideInputBuffer(void* state)
{   /* lz4.c:2813: the reference moves the last 64 KB to the buffer's start; the buffer is not tracked here, only the dictionary's end */
    LZ4_stream_t* s = (LZ4_stream_t*)state;
    return (char*)(uintptr_t)(s->internal_donotuse.dictionary + s->internal_donotuse.dictSize);
}
int LZ4_compressHC(const char* src, char* dst, int srcSize) { return LZ4_compress_HC(src, dst, srcSize, LZ4_compressBound(srcSize), 0); }
int LZ4_compressHC_limitedOutput(const char* src, char* dst, int srcSize, int maxDstSize) { return LZ4_compress_HC(src, dst, srcSize, maxDstSize, 0); }
int LZ4_compressHC2(const char* src, char* dst, int srcSize, int cLevel) { return LZ4_compress_HC(src, dst, srcSize, LZ4_compressBound(srcSize), cLevel); }
int LZ4_compressHC2_limitedOutput(const char* src, char* dst, int srcSize, int maxDstSize, int cLevel) { return LZ4_compress_HC(src, dst, srcSize, maxDstSize, cLevel); }
int LZ4_compressHC_withStateHC(void* state, const char* src, char* dst, int srcSize) { return LZ4_compress_HC_extStateHC(state, src, dst, srcSize, LZ4_compressBound(srcSize), 0); }
int LZ4_compressHC_limitedOutput_withStateHC(void* state, const char* src, char* dst, int srcSize, int maxDstSize) { return LZ4_compress_HC_extStateHC(state, src, dst, srcSize, maxDstSize, 0); }
int LZ4_compressHC2_withStateHC(void* state, const char* src, char* dst, int srcSize, int cLevel) { return LZ4_compress_HC_extStateHC(state, src, dst, srcSize, LZ4_compressBound(srcSize), cLevel); }
int LZ4_compressHC2_limitedOutput_withStateHC(void* state, const char* src, char* dst, int srcSize, int maxDstSize, int cLevel) { return LZ4_compress_HC_extStateHC(state, src, dst, srcSize, maxDstSize, cLevel); }
int LZ4_compressHC_continue(LZ4_streamHC_t* s, const char* src, char* dst, int srcSize) { return LZ4_compress_HC_continue(s, src, dst, srcSize, LZ4_compressBound(srcSize)); }
int LZ4_compressHC_limitedOutput_continue(LZ4_streamHC_t* s, const char* src, char* dst, int srcSize, int maxDstSize) { return LZ4_compress_HC_continue(s, src, dst, srcSize, maxDstSize); }
int LZ4_sizeofStreamStateHC(void) { return (int)sizeof(LZ4_streamHC_t); }
int LZ4_resetStreamStateHC(void* state, char* inputBuffer) { (void)inputBuffer; return LZ4_initStreamHC(state, sizeof(LZ4_streamHC_t)) ? 0 : 1; }
void* LZ4_createHC(const char* inputBuffer) { (void)inputBuffer; return LZ4_createStreamHC(); }
int LZ4_freeHC(void* LZ4HC_Data) { return LZ4_freeStreamHC((LZ4_streamHC_t*)LZ4HC_Data); }
int LZ4_compressHC2_continue(void* LZ4HC_Data, const char* src, char* dst, int srcSize, int cLevel)
{   LZ4_setCompressionLevel((LZ4_streamHC_t*)LZ4HC_Data, cLevel); return LZ4_compress_HC_continue((LZ4_streamHC_t*)LZ4HC_Data, src, dst, srcSize, LZ4_compressBound(srcSize)); }
int LZ4_compressHC2_limitedOutput_continue(void* LZ4HC_Data, const char* src, char* dst, int srcSize, int maxDstSize, int cLevel)
{   LZ4_setCompressionLevel((LZ4_streamHC_t*)LZ4HC_Data, cLevel); return LZ4_compress_HC_continue((LZ4_streamHC_t*)LZ4HC_Data, src, dst, srcSize, maxDstSize); }
char* LZ4_slideInputBufferHC(void* LZ4HC_Data)
{   LZ4_streamHC_t* s = (LZ4_streamHC_t*)LZ4HC_Data; return (char*)(uintptr_t)(s->internal_donotuse.dictionary + s->internal_donotuse.dictSize); }

/* ------------------------------------------------------------------ (3) xxHash, exported as the reference's shared library does
 * (XXH_NAMESPACE=LZ4_).  XXH32: csrc/xxh32_host.h.  XXH64: from the xxHash specification (doc/xxhash_spec.md of xxHash). */
typedef struct { uint32_t total_len_32, large_len, v[4], mem32[4], memsize, reserved; } LZ4_XXH32_state_t;    /* xxhash.h:XXH32_state_s */
typedef struct { unsigned char digest[4]; } LZ4_XXH32_canonical_t;
unsigned LZ4_XXH_versionNumber(void) { return 803; }                /* the xxHash release the reference bundles (0.8.3) */
uint32_t LZ4_XXH32(const void* input, size_t len, uint32_t seed)
{
    xxh32_state s; xxh32_reset_seed(&s, seed); xxh32_update(&s, (const uint8_t*)input, len); return xxh32_digest(&s);
}
LZ4_XXH32_state_t* LZ4_XXH32_createState(void) { return (LZ4_XXH32_state_t*)calloc(1, sizeof(LZ4_XXH32_state_t) > sizeof(xxh32_state) ? sizeof(LZ4_XXH32_state_t) : sizeof(xxh32_state)); }
int LZ4_XXH32_freeState(LZ4_XXH32_state_t* p) { free(p); return 0; }
void LZ4_XXH32_copyState(LZ4_XXH32_state_t* d, const LZ4_XXH32_state_t* s) { memcpy(d, s, sizeof *d); }
int LZ4_XXH32_reset(LZ4_XXH32_state_t* p, uint32_t seed) { if (!p) return 1; xxh32_reset_seed((xxh32_state*)p, seed); return 0; }
int LZ4_XXH32_update(LZ4_XXH32_state_t* p, const void* in, size_t len) { if (!p) return 1; if (in && len) xxh32_update((xxh32_state*)p, (const uint8_t*)in, len); return 0; }
uint32_t LZ4_XXH32_digest(const LZ4_XXH32_state_t* p) { return xxh32_digest((const xxh32_state*)p); }
void LZ4_XXH32_canonicalFromHash(LZ4_XXH32_canonical_t* d, uint32_t h) { d->digest[0] = (unsigned char)(h >> 24); d->digest[1] = (unsigned char)(h >> 16); d->digest[2] = (unsigned char)(h >> 8); d->digest[3] = (unsigned char)h; }
uint32_t LZ4_XXH32_hashFromCanonical(const LZ4_XXH32_canonical_t* s) { return ((uint32_t)s->digest[0] << 24) | ((uint32_t)s->digest[1] << 16) | ((uint32_t)s->digest[2] << 8) | s->digest[3]; }

#define P64_1 0x9E3779B185EBCA87ull
#define P64_2 0xC2B2AE3D27D4EB4Full
#define P64_3 0x165667B19E3779F9ull
#define P64_4 0x85EBCA77C2B2AE63ull
#define P64_5 0x27D4EB2F165667C5ull
typedef struct { uint64_t total_len, v[4], mem64[4]; uint32_t memsize, reserved32; uint64_t reserved64; } LZ4_XXH64_state_t;   /* xxhash.h:XXH64_state_s */
typedef struct { unsigned char digest[8]; } LZ4_XXH64_canonical_t;
static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t rd64le(const uint8_t* p) { uint64_t v = 0; int i; for (i = 7; i >= 0; i--) v = (v << 8) | p[i]; return v; }
static uint32_t rd32le(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t xxh64_round(uint64_t acc, uint64_t in) { acc += in * P64_2; acc = rotl64(acc, 31); return acc * P64_1; }
static uint64_t xxh64_merge(uint64_t h, uint64_t v) { h ^= xxh64_round(0, v); return h * P64_1 + P64_4; }
int LZ4_XXH64_reset(LZ4_XXH64_state_t* s, uint64_t seed)
{
    if (!s) return 1;
    memset(s, 0, sizeof *s);
    s->v[0] = seed + P64_1 + P64_2; s->v[1] = seed + P64_2; s->v[2] = seed; s->v[3] = seed - P64_1;
    return 0;
}
int LZ4_XXH64_update(LZ4_XXH64_state_t* s, const void* input, size_t len)
{
    const uint8_t* p = (const uint8_t*)input;
    if (!s) return 1;
    if (!p || !len) return 0;
    s->total_len += len;
    if (s->memsize + len < 32) { memcpy((uint8_t*)s->mem64 + s->memsize, p, len); s->memsize += (uint32_t)len; return 0; }
    if (s->memsize) {
        const size_t fill = 32 - s->memsize;
        int i;
        memcpy((uint8_t*)s->mem64 + s->memsize, p, fill);
        for (i = 0; i < 4; i++) s->v[i] = xxh64_round(s->v[i], rd64le((const uint8_t*)s->mem64 + 8 * i));
        p += fill; len -= fill; s->memsize = 0;
    }
    while (len >= 32) { int i; for (i = 0; i < 4; i++) s->v[i] = xxh64_round(s->v[i], rd64le(p + 8 * i)); p += 32; len -= 32; }
    if (len) { memcpy(s->mem64, p, len); s->memsize = (uint32_t)len; }
    return 0;
}
uint64_t LZ4_XXH64_digest(const LZ4_XXH64_state_t* s)
{
    const uint8_t* p = (const uint8_t*)s->mem64;
    size_t len = s->memsize;
    uint64_t h;
    if (s->total_len >= 32) {
        h = rotl64(s->v[0], 1) + rotl64(s->v[1], 7) + rotl64(s->v[2], 12) + rotl64(s->v[3], 18);
        h = xxh64_merge(h, s->v[0]); h = xxh64_merge(h, s->v[1]); h = xxh64_merge(h, s->v[2]); h = xxh64_merge(h, s->v[3]);
    } else h = s->v[2] + P64_5;
    h += s->total_len;
    while (len >= 8) { h ^= xxh64_round(0, rd64le(p)); h = rotl64(h, 27) * P64_1 + P64_4; p += 8; len -= 8; }
    if (len >= 4) { h ^= (uint64_t)rd32le(p) * P64_1; h = rotl64(h, 23) * P64_2 + P64_3; p += 4; len -= 4; }
    while (len) { h ^= (*p) * P64_5; h = rotl64(h, 11) * P64_1; p++; len--; }
    h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; h ^= h >> 32;
    return h;
}
uint64_t LZ4_XXH64(const void* input, size_t len, uint64_t seed)
{
    LZ4_XXH64_state_t s; LZ4_XXH64_reset(&s, seed); LZ4_XXH64_update(&s, input, len); return LZ4_XXH64_digest(&s);
}
LZ4_XXH64_state_t* LZ4_XXH64_createState(void) { return (LZ4_XXH64_state_t*)calloc(1, sizeof(LZ4_XXH64_state_t)); }
int LZ4_XXH64_freeState(LZ4_XXH64_state_t* p) { free(p); return 0; }
void LZ4_XXH64_copyState(LZ4_XXH64_state_t* d, const LZ4_XXH64_state_t* s) { memcpy(d, s, sizeof *d); }
void LZ4_XXH64_canonicalFromHash(LZ4_XXH64_canonical_t* d, uint64_t h) { int i; for (i = 0; i < 8; i++) d->digest[i] = (unsigned char)(h >> (56 - 8 * i)); }
uint64_t LZ4_XXH64_hashFromCanonical(const LZ4_XXH64_canonical_t* s) { uint64_t h = 0; int i; for (i = 0; i < 8; i++) h = (h << 8) | s->digest[i]; return h; }
