// chain_spec_kernel.h -- linked blocks (lz4frame.c:1901-1915: a block's matches reach 64 KB back into the output of the blocks before it)
// decoded side by side, for gfx950.
//
// A byte of LZ4 output is a copy of exactly ONE earlier byte: a literal of its block, or - through a chain of matches - a byte of the history
// the block started with.  So a stretch of the chain can be decoded before its history exists: decode it against a made-up history and every
// byte that does not come from the history is already right; for the others it is enough to know WHICH history byte they copy.  The chain is cut
// in UNITS (one block of 1 MiB and more; as many smaller blocks as make 1 MiB).  Every unit but the first is decoded by the ordinary
// decoder - lz4amd_k_decompress, ONE launch: a unit's blocks are a run of dependent blocks of that launch, the runs do not wait for each other
// (lz4amd_dec_params.chain) - against made-up histories whose byte i (lo = i & 0xFF, hi = i >> 8) is
//     A: lo        B: lo + (255 - hi) + 1 (mod 256)        C: ~lo
// A and B differ wherever hi != 0, and then {A, B} give the index back: an output byte depends on the history exactly when its A and B values
// differ - unless it copies one of the FIRST 256 bytes of the 64 KB (hi = 0: B = A).  Only matches in a unit's first 255 bytes with offsets
// above 65280 read those; every block reports whether it has one (lz4amd_dec_params.lowref), and only such units are decoded a third
// time, against C, which differs from A in every byte: the run of variant C is GATED on what the first block of variant A reports
// (lz4amd_dec_params.gate; tickets: A's blocks, then C's, then B's - a run that is not wanted costs a look at a flag).  A unit whose first
// block is so small that a later one could begin within those 255 bytes is decoded three times without asking.  Chains of large blocks (a unit = one
// block): A's decode writes the block's entry-point table on its way (lz4amd_dec_params.hint_make), and B is decoded by a SECOND launch - the
// ordinary kernel over independent blocks with 64 KB in front - from those tables, without the decoder's first stage (spec_result_b).  (Two histories cannot
// tell all 65 536 positions and "no history" apart: two bytes that differ are 65 280 pairs.)  What follows is bandwidth work:
//   spec_scan     sizes -> output positions (the decoded sizes do not depend on the history's content); the first unit with a bad block; which
//                 units were decoded against C
//   spec_merge    all units at once: A's bytes go to their place where A == B (== C); the last byte that depends on the history is noted per unit
//   spec_patch    per unit, bytes 0 .. last dependent: out[p] = out[unit start - 65536 + index].  The history is complete by then, except where it
//                 overlaps a unit's patched stretch - only then a unit waits for the unit before it (always, for units of less than 64 KB).
//                 A reference before the start of the data (lz4.c:2356) is an error, found here.
//   spec_results  per block: the decoded sizes up to the first failure, -1 from there on (a linked frame ends at its first bad block)
// The serial part of a linked frame shrinks from every block's copy stage (~1.2 ms per 4 MiB block on one CU) to a unit's own blocks and the
// stretches that really depend on the unit before: on 4 MiB blocks of datagen -P60 the first ~100 KB of a block, patched without waiting.
#pragma once
#include "lz4_common.h"
#include "../lz4amd_params.h"

namespace lz4amd {

using SpecBatch = ::lz4amd_spec_params;
constexpr uint32_t kSpecHist = 65536, kSpecSlice = 65536, kSpecThreads = 256, kSpecScanThreads = 1024, kSpecPatchThreads = 1024, kSpecPre = kSpecHist / (kSpecPatchThreads * 8), kSpecFillParts = kSpecHist / (kSpecThreads * 16);

__device__ __forceinline__ uint8_t* spec_slot(const SpecBatch& P, uint32_t u, uint32_t v) {       // unit u >= 1, variant v: its made-up history, then its output
    return P.slots + ((uint64_t)(u - 1) * 3 + v) * P.slot_stride;
}
__device__ __forceinline__ uint32_t spec_byte(const lz4amd_u32x4& v, uint32_t j) { return (v[j >> 2] >> ((j & 3) * 8)) & 0xFFu; }
__device__ __forceinline__ uint32_t spec_len(const SpecBatch& P, uint32_t u) { const uint32_t r = P.n - u * P.group; return r < P.group ? r : P.group; }      // blocks of unit u
__device__ __forceinline__ uint32_t spec_entry(const SpecBatch& P, uint32_t u, uint32_t v, uint32_t j) {      // where lz4amd_k_decompress left the result of block j of unit u, variant v (0: A, 1: B, 2: C)
    return u == 0 ? j : spec_len(P, 0) + (u - 1) * 3 * P.group + v * spec_len(P, u) + j;
}
// decoded size of block j of unit u, -1 if one of its decodes failed (they fail alike: what a block refers to does not depend on the history's content)
__device__ __forceinline__ int32_t spec_block_size(const SpecBatch& P, uint32_t u, uint32_t j) {
    const int32_t a = P.spec_result[spec_entry(P, u, 0, j)];
    if (u == 0) return a;
    const int32_t b = P.spec_result_b ? P.spec_result_b[2 * (u - 1)] : P.spec_result[spec_entry(P, u, 1, j)];      // (a second launch: units of one block)
    return (a >= 0 && a == b) ? a : -1;
}
// ---- the made-up histories.  grid: 3 * (n_units - 1) * kSpecFillParts workgroups of kSpecThreads
__device__ __forceinline__ void spec_fill_body(const SpecBatch& P) {
    const uint32_t e = blockIdx.x / kSpecFillParts, part = blockIdx.x % kSpecFillParts, v = e % 3;
    uint8_t* base = P.slots + (uint64_t)e * P.slot_stride;
    const uint32_t i0 = part * (kSpecThreads * 16) + threadIdx.x * 16;
    lz4amd_u32x4 w;
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
        uint32_t x = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t i = i0 + q * 4 + j;
            const uint32_t b = v == 0 ? (i & 0xFFu) : v == 1 ? ((i & 0xFFu) + (255u - (i >> 8)) + 1u) & 0xFFu : (~i & 0xFFu);
            x |= b << (j * 8);
        }
        w[q] = x;
    }
    st_global16_raw(base + i0, w);
}

// ---- chains of large blocks, between the two launches: unit u's table (A's decode wrote it) goes to the entries of its B and C in the second
//      launch; C is wanted only if A's block reports a match that reads the first 256 bytes of the 64 KB (source size 0: the block answers -1 at once).
//      grid: n_units - 1 workgroups of kSpecThreads
__device__ __forceinline__ void spec_gate_body(const SpecBatch& P) {
    const uint32_t u = blockIdx.x + 1, tid = threadIdx.x;
    const uint32_t a = spec_entry(P, u, 0, 0);
    const bool three = P.lowref[a] == 1u;
    const lz4amd_u32x4* src = (const lz4amd_u32x4*)(P.tables_a + (uint64_t)a * P.table_stride);
    lz4amd_u32x4* db = (lz4amd_u32x4*)(P.tables_b + (uint64_t)(2 * (u - 1)) * P.table_stride);
    lz4amd_u32x4* dc = (lz4amd_u32x4*)(P.tables_b + (uint64_t)(2 * (u - 1) + 1) * P.table_stride);
    const lz4amd_u32x4 h0 = src[0], h1 = src[1];                      // { magic, output bytes, compressed bytes, sequences }, { rows, 0, 0, 0 }
    uint64_t bytes = 0;                                                // (no table: the copies say so too)
    if (h0[0] == LZ4AMD_HINT_MAGIC && LZ4AMD_HINT_HEAD + ((uint64_t)h1[0] + 1) * LZ4AMD_HINT_ROW <= P.table_stride) bytes = LZ4AMD_HINT_HEAD + ((uint64_t)h1[0] + 1) * LZ4AMD_HINT_ROW;
    const uint32_t n16 = (uint32_t)((bytes + 15) / 16);
    if (n16 == 0) { if (tid == 0) { lz4amd_u32x4 z = {0, 0, 0, 0}; db[0] = z; dc[0] = z; } }
    for (uint32_t i = tid; i < n16; i += kSpecThreads) { const lz4amd_u32x4 v = src[i]; db[i] = v; if (three) dc[i] = v; }
    if (tid == 0) { P.b_src_size[2 * (u - 1)] = P.src_size[u]; P.b_src_size[2 * (u - 1) + 1] = three ? P.src_size[u] : 0; }
}

// ---- sizes and positions.  ONE workgroup of kSpecScanThreads
__device__ __forceinline__ uint32_t spec_unit_size(const SpecBatch& P, uint32_t u, bool& whole) {      // bytes of the unit's blocks up to its first bad one
    uint32_t sum = 0;
    whole = true;
    for (uint32_t j = 0, m = spec_len(P, u); j < m; j++) {
        const int32_t s = spec_block_size(P, u, j);
        if (s < 0) { whole = false; break; }
        sum += (uint32_t)s;
    }
    return sum;
}
__device__ __forceinline__ void spec_scan_body(const SpecBatch& P) {
    __shared__ unsigned long long sums[2][kSpecScanThreads];
    __shared__ uint32_t first_bad;
    const uint32_t t = threadIdx.x, n = P.n_units;
    const uint32_t per = (n + kSpecScanThreads - 1) / kSpecScanThreads;
    const uint32_t k0 = t * per < n ? t * per : n, k1 = k0 + per < n ? k0 + per : n;
    if (t == 0) first_bad = n;
    __syncthreads();
    unsigned long long mine = 0;
    for (uint32_t k = k0; k < k1; k++) {
        bool whole;
        mine += spec_unit_size(P, k, whole);
        if (!whole) { atomicMin(&first_bad, k); break; }
    }
    sums[0][t] = mine;
    __syncthreads();
    uint32_t cur = 0;
    for (uint32_t d = 1; d < kSpecScanThreads; d <<= 1) {                 // inclusive scan of the threads' sums
        sums[cur ^ 1][t] = sums[cur][t] + (t >= d ? sums[cur][t - d] : 0ull);
        cur ^= 1;
        __syncthreads();
    }
    unsigned long long at = sums[cur][t] - mine;
    const uint32_t bad = first_bad;
    for (uint32_t k = k0; k < k1; k++) {
        bool whole;
        const uint32_t s = k <= bad ? spec_unit_size(P, k, whole) : 0u;   // (the unit with the bad block still has its good blocks put in place)
        P.start[k] = (long long)at; P.size[k] = (int32_t)s;
        P.lastdep[k] = -1; P.done[k] = 0; P.badpos[k] = 0x7FFFFFFF;
        at += s;
        bool three = false;                                                // was the unit decoded against C?  (a block reports 1 or 0, whatever becomes of it)
        if (k >= 1 && k <= bad) for (uint32_t j = 0, m = spec_len(P, k); j < m; j++) three = three || P.lowref[spec_entry(P, k, 0, j)] == 1u;
        P.three[k] = three ? 1 : 0;
    }
    if (t == 0) { P.info[0] = bad < n ? bad + 1 : n; P.info[1] = 0; }
}

// ---- bytes that do not depend on the history go to their place.  grid: (n_units, slices of kSpecSlice), kSpecThreads
__device__ __forceinline__ void spec_merge_body(const SpecBatch& P) {
    __shared__ int32_t wg_dep;
    const uint32_t k = blockIdx.x, tid = threadIdx.x;
    if (k == 0 || k >= P.info[0]) return;                                  // (unit 0 is decoded in place, against the history that is really there)
    const uint32_t size = (uint32_t)P.size[k], lo = blockIdx.y * kSpecSlice;
    if (lo >= size) return;
    const uint32_t hi = lo + kSpecSlice < size ? lo + kSpecSlice : size;
    uint8_t* out = P.out + P.start[k];
    if (tid == 0) wg_dep = -1;
    __syncthreads();
    const uint8_t* A = spec_slot(P, k, 0) + kSpecHist;
    const uint8_t* C = spec_slot(P, k, P.three[k] ? 2 : 1) + kSpecHist;       // what tells the bytes that come from the history: B, or C if B cannot
    const bool aligned = (((uintptr_t)out) & 15u) == 0;
    int32_t dep = -1;
    for (uint32_t p = lo + tid * 16; p < hi; p += kSpecThreads * 16) {
        const lz4amd_u32x4 a = ld_global16_raw(A + p), c = ld_global16_raw(C + p);
        const uint32_t nb = hi - p < 16 ? hi - p : 16;
        const bool same = ((a[0] ^ c[0]) | (a[1] ^ c[1]) | (a[2] ^ c[2]) | (a[3] ^ c[3])) == 0;
        if (same && nb == 16 && aligned) st_global16_raw(out + p, a);
        else
            for (uint32_t j = 0; j < nb; j++) {
                const uint32_t x = spec_byte(a, j);
                if (x == spec_byte(c, j)) out[p + j] = (uint8_t)x; else dep = (int32_t)(p + j);
            }
    }
    if (dep >= 0) atomicMax(&wg_dep, dep);
    __syncthreads();
    if (tid == 0 && wg_dep >= 0) atomicMax(&P.lastdep[k], wg_dep);
}

// ---- bytes that do.  Workgroups of kSpecPatchThreads take blocks in order from a ticket counter (a block that has to wait for the one before
//      finds it taken by a running workgroup; "block" below: a unit).  A thread gathers the history bytes of its 16 (sixteen loads in flight, always from before the
//      block: no store of the block can alias them) and stores them with A's bytes as one piece.
constexpr uint32_t kSpecPatchLds = kSpecHist + 32;                        // the unit's history, from the 8-byte boundary below its first byte
__device__ __forceinline__ void spec_patch_body(const SpecBatch& P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t spec_hist[];
    __shared__ uint32_t s_k, s_bad;
    const uint32_t tid = threadIdx.x, n = P.n_units, nvalid = P.info[0];
    for (;;) {
        __syncthreads();
        if (tid == 0) { s_k = take_ticket(&P.info[1]); s_bad = 0x7FFFFFFFu; }
        __syncthreads();
        const uint32_t k = s_k;
        if (k >= n) break;
        const bool live = k >= 1 && k < nvalid;
        const int32_t last = live ? P.lastdep[k] : -1;
        const long long start = P.start[k];
        const uint8_t* A = live ? spec_slot(P, k, 0) + kSpecHist : nullptr;
        const uint8_t* B = live ? spec_slot(P, k, 1) + kSpecHist : nullptr;
        const uint8_t* C = live ? spec_slot(P, k, P.three[k] ? 2 : 1) + kSpecHist : nullptr;
        // A thread works on batches of kSpecPre pieces of FOUR bytes (a batch of the workgroup: 32 KB - sixteen pieces a thread spilled registers): all of a batch's loads from the slots first
        // (the compiler cannot know that the stores to the output never hit what the next piece reads); the history bytes come out of the LDS.
        // What does not depend on the block before - the first batch's loads - is read before waiting for it.
        uint32_t pa[kSpecPre], pb[kSpecPre], px[kSpecPre];
        auto load_batch = [&](uint32_t first) {
#pragma unroll
            for (uint32_t i = 0; i < kSpecPre; i++) {
                const uint32_t p = ((first + i) * kSpecPatchThreads + tid) * 4;
                if (last >= 0 && p <= (uint32_t)last) {
                    pa[i] = *(const uint32_t*)(A + p); pb[i] = *(const uint32_t*)(B + p);
                    px[i] = pa[i] ^ *(const uint32_t*)(C + p);
                } else px[i] = 0;
            }
        };
        load_batch(0);
        // history that may still change: the patched stretch of the block before, if it reaches into my 64 KB; blocks of less than 64 KB
        // pass the question on (they publish only after the block before them has)
        if (k >= 1 && k < nvalid) {
            const bool small_prev = (uint32_t)P.size[k - 1] < kSpecHist, small_me = (uint32_t)P.size[k] < kSpecHist;
            const int32_t lp = P.lastdep[k - 1];
            const bool prev_reaches = small_prev || (lp >= 0 && P.start[k - 1] + lp >= start - (long long)kSpecHist);
            if ((last >= 0 && prev_reaches) || small_me) {
                // (no acquire / release fences in this kernel: a fence at device scope writes back and empties the caches - tens of microseconds per
                //  hand-over with sixteen waves doing it.  What one workgroup hands to another - the patched bytes, the word that says so - is
                //  written and read with device-scope accesses instead, which go past the caches that are not shared; the word is written once
                //  every store of the workgroup has been acknowledged.)
                if (tid == 0) while (__hip_atomic_load(&P.done[k - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
                __syncthreads();
            }
        }
        if (last >= 0) {
            uint8_t* out = P.out + start;
            const uint8_t* hist = out - kSpecHist;                          // made-up history byte i stands for hist[i]
            const uint32_t before = start + (long long)P.prefix0 < (long long)kSpecHist ? (uint32_t)(start + (long long)P.prefix0) : kSpecHist;      // bytes of data in front of the block, 64 KB of them count
            const bool aligned = (((uintptr_t)out) & 3u) == 0;
            uint32_t bad = 0x7FFFFFFFu;
            // the history goes to the LDS in 8-byte pieces (lanes side by side), the gathers read it there: a gather from memory costs a cache
            // line per lane - ~100 cycles per wave instruction, 40 us per 64 KB of patched bytes on one CU
            const uint32_t shift = (uint32_t)((uintptr_t)hist & 7u);
            {
                const unsigned long long* g = (const unsigned long long*)(hist - shift);
                const uint32_t first = (kSpecHist - before + shift) / 8;     // (the piece that holds the first byte of data; nothing is read below it)
                for (uint32_t i = first + tid; i < (kSpecHist + shift + 7) / 8; i += kSpecPatchThreads)
                    ((unsigned long long*)spec_hist)[i] = __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            for (uint32_t first = 0; first * kSpecPatchThreads * 4 <= (uint32_t)last; first += kSpecPre) {
                if (first) load_batch(first);
#pragma unroll
                for (uint32_t i = 0; i < kSpecPre; i++) {
                    const uint32_t p = ((first + i) * kSpecPatchThreads + tid) * 4, a = pa[i], b = pb[i], x = px[i];
                    const uint32_t nb = x == 0 ? 0u : (uint32_t)last + 1 - p < 4 ? (uint32_t)last + 1 - p : 4;
                    uint32_t w = 0;
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++) {
                        const uint32_t aj = (a >> (8 * j)) & 0xFFu, idx = ((255u - ((((b >> (8 * j)) & 0xFFu) - aj - 1u) & 0xFFu)) << 8) | aj;
                        const bool dep = ((x >> (8 * j)) & 0xFFu) != 0 && j < nb;
                        if (dep && kSpecHist - idx > before && p + j < bad) bad = p + j;       // lz4.c:2356: before the start of the data
                        const bool take = dep && kSpecHist - idx <= before;
                        const uint32_t hv = spec_hist[take ? shift + idx : 0u];      // (no branch around the read: a read in a branch of its own is waited for there)
                        w |= (take ? hv : aj) << (8 * j);
                    }
                    if (x == 0) continue;
                    if (aligned && nb == 4) __hip_atomic_store((uint32_t*)(out + p), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        for (uint32_t j = 0; j < nb; j++)
                            if ((x >> (8 * j)) & 0xFFu) __hip_atomic_store(out + p + j, (uint8_t)(w >> (8 * j)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (bad != 0x7FFFFFFFu) atomicMin(&s_bad, bad);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // my stores have arrived
        __syncthreads();
        if (tid == 0) {
            if (s_bad != 0x7FFFFFFFu) P.badpos[k] = (int32_t)s_bad;
            __hip_atomic_store(&P.done[k], 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- results, per block.  ONE workgroup of kSpecScanThreads
__device__ __forceinline__ void spec_results_body(const SpecBatch& P) {
    __shared__ uint32_t first_bad;
    const uint32_t t = threadIdx.x, n = P.n, G = P.group;
    if (t == 0) first_bad = n;
    __syncthreads();
    for (uint32_t b = t; b < n; b += kSpecScanThreads) {
        const uint32_t u = b / G, j = b % G;
        const int32_t sz = spec_block_size(P, u, j);
        // (a unit's third decode answers like the other two)
        if (sz < 0 || (u >= 1 && u < P.info[0] && P.three[u] && (P.spec_result_b ? P.spec_result_b[2 * (u - 1) + 1] : P.spec_result[spec_entry(P, u, 2, j)]) != sz)) atomicMin(&first_bad, b);
    }
    for (uint32_t u = t; u < P.n_units; u += kSpecScanThreads) {
        const int32_t bp = P.badpos[u];
        if (bp == 0x7FFFFFFF) continue;
        uint32_t at = 0, j = 0;                                             // the block that holds the unit's first bad byte
        for (const uint32_t m = spec_len(P, u); j + 1 < m; j++) {
            const int32_t s = spec_block_size(P, u, j);
            if (s < 0 || at + (uint32_t)s > (uint32_t)bp) break;
            at += (uint32_t)s;
        }
        atomicMin(&first_bad, u * G + j);
    }
    __syncthreads();
    for (uint32_t b = t; b < n; b += kSpecScanThreads) P.result[b] = b < first_bad ? spec_block_size(P, b / G, b % G) : -1;
}

} // namespace lz4amd
