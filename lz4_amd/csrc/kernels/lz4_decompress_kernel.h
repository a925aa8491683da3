// lz4_decompress_kernel.h -- batched LZ4 block decompression for gfx950 (MI355X).
//
// Replaces, for a whole batch of independent blocks resident in HBM, what the reference does
// per block in LZ4_decompress_safe (lib/lz4.c:2451 -> LZ4_decompress_generic lz4.c:2023-2445;
// length fields: read_variable_length lz4.c:1979-2014; end-of-block rules lz4.c:2276-2330,
// 2421-2429).  Accepts ANY legal LZ4 block (reference-produced included), rejects what the
// reference's safe loop rejects, never reads outside src[0,csize) nor writes outside
// dst[0,cap).  This is not a port: the reference decoder is one serial token chain per block;
// here ONE 1024-thread workgroup (16 waves, one CU) decodes a block in five phases:
//
//   1 WALK    the compressed stream is cut in <=1024 segments; every thread follows the token
//             chain of its own segment, starting kWarm bytes EARLY at an arbitrary byte and
//             relying on LZ4 chains self-synchronising (a wrong start merges with the true
//             chain after a few hundred bytes).  Only token/length bytes are read.
//   2 FIX     fix-point: segment j is right iff it started where segment j-1 exited; threads
//             whose guess was wrong re-walk from the true entry.  Segment 0 starts at byte 0,
//             so by induction the result is the true chain for every input (worst case: 1024
//             iterations for adversarial streams; 1-2 on real data).
//   3 SCAN    block-wide prefix sums of per-segment token counts / output bytes.
//   4 EMIT    every thread re-walks its segment from the true entry and writes one 16-byte
//             record per sequence (output position, literal source, literal length, offset)
//             into an L2-resident table, applying the output-side format rules, plus a coarse
//             index: for every 1 KB REGION of the output, the sequence that covers its first
//             byte.
//   5 COPY    output-stationary and barrier-free.  The output is cut in 1 KB regions; wave w
//             owns regions w, w+16, w+32, ... and lane k of the wave owns the 16-byte CHUNK k
//             of the region.  A lane composes its chunk in four VGPRs: literal pieces are
//             unaligned 16-byte loads straight from the compressed stream, match pieces are
//             unaligned 16-byte reads from a 128 KB LDS ring that always holds the 64 KB LZ4
//             window; finished chunks go to the ring (one ds_write_b128) and to HBM (one
//             16-byte store per lane, 1 KB contiguous per wave).  Match sources that are not
//             final yet are waited for through per-chunk done bits (one u64 per region, single
//             writer) and per-wave completed-region counters in LDS - pure dataflow between
//             the 16 waves, no __syncthreads() inside the phase.  Dependencies always point to
//             lower output positions and every wave walks its regions in increasing order, so
//             the lowest unfinished region can always finish: no deadlock.  A wave may run at
//             most 63 regions ahead of the slowest one (the ring slot it overwrites must be
//             out of every other wave's window).
//
// HBM traffic per block: compressed bytes read ~2x (walk reads tokens only; copy reads the
// literals), sequence table written+read once (16 B per sequence), output written once,
// matches never touch HBM.  No MFMA: this is byte shuffling.
#pragma once
#include "lz4_common.h"
#include "../lz4amd_params.h"

namespace lz4amd {

using DecBatch = ::lz4amd_dec_params;     // argument block (lz4amd_params.h)

struct alignas(16) SeqRec { uint32_t outpos, litpos, ll, off; };

enum : uint32_t {
    kDecThreads = 1024,
    kDecWaves = kDecThreads / 64,
    kRingBytes = 128u << 10,
    kRingMask = kRingBytes - 1,
    kChunk = 16,                                // output bytes owned by one lane per region
    kRegionShift = 10,
    kRegion = 1u << kRegionShift,               // 64 lanes x 16 bytes
    kSlots = kRingBytes / kRegion,              // ring slots (regions)
    kMaxLead = 63,                              // a wave may lead the slowest by this many regions
    kWarm = 2048,                               // speculative warm-up distance
    kMinSeg = 2048,                             // minimum segment length
    kNone = 0xFFFFFFFFu,
};

// LDS carve-up (bytes)
enum : uint32_t {
    kOffRing = 0,
    kOffFirst = kOffRing + kRingBytes,                       // u32[kDecWaves][64]
    kOffBits = kOffFirst + kDecWaves * 64 * 4,               // u64[kSlots] per-chunk done bits
    kOffFin = kOffBits + kSlots * 8,                         // u32[kDecWaves] regions completed
    kOffScan = kOffFin + kDecWaves * 4,                      // u32[64] (3 per wave needed)
    kOffMisc = kOffScan + 64 * 4,                            // u32[16]
    kDecLdsBytes = kOffMisc + 16 * 4,
    // phase 1-4 arrays overlay the ring (not live at the same time)
    kOffSegEntry = 0,
    kOffSegExit = kOffSegEntry + kDecThreads * 4,
    kOffSegN = kOffSegExit + kDecThreads * 4,
    kOffSegOb = kOffSegN + kDecThreads * 4,
};
enum : uint32_t { M_BLOCK = 0, M_ERR = 1 };

// scratch of one workgroup: sequence table, then the region index
__host__ __device__ inline uint64_t dec_table_bytes(uint32_t max_csize) {
    // every sequence but the last takes >= 3 compressed bytes; +1 last, +1 sentinel
    return ((uint64_t)max_csize / 3 + 4) * sizeof(SeqRec);
}
__host__ __device__ inline uint64_t dec_scratch_bytes(uint32_t max_csize, uint32_t max_cap) {
    return dec_table_bytes(max_csize) + (((uint64_t)max_cap >> kRegionShift) + 4) * 4;
}

struct WalkOut { uint32_t exit, n, ob, err; };

// Follow the token chain from p while p < e (e <= csize).  err != 0 => malformed at err-1.
// EMIT: also write SeqRec's from index `seq` / output position `o`, fill the region index, and
// apply the output-side rules (needs cap).  The input-side rules are those of the reference's
// safe loop.
template <bool EMIT>
__device__ __forceinline__ WalkOut walk_chain(const uint8_t* __restrict__ src, uint32_t csize,
                                              uint32_t p, uint32_t e, SeqRec* __restrict__ tab,
                                              uint32_t* __restrict__ idx,
                                              uint32_t seq, uint32_t o, uint32_t cap) {
    WalkOut r; r.n = 0; r.ob = 0; r.err = 0;
    while (p < e) {
        const uint32_t t = ld_u8(src + p);
        uint32_t ll = t >> 4;
        uint32_t q = p + 1;
        if (ll == 15) {                          // lz4.c:1979-2014, limit iend-15
            uint32_t b;
            do {
                if (q + 15 >= csize) { r.err = p + 1; break; }
                b = ld_u8(src + q); q++; ll += b;
                if (ll > csize) { r.err = p + 1; break; }
            } while (b == 255);
            if (r.err) break;
        }
        const uint32_t rem = csize - q;          // q <= csize always holds here
        bool last = (rem < ll + 8);              // lz4.c:2279 input-side restriction
        if (EMIT) last = last || (cap - o < ll + kMfLimit);      // output-side restriction
        if (last) {
            if (rem != ll) { r.err = p + 1; break; }             // must end the input exactly
            if (EMIT) {
                if (cap - o < ll) { r.err = p + 1; break; }
                SeqRec rec; rec.outpos = o; rec.litpos = q; rec.ll = ll; rec.off = 0;
                tab[seq] = rec;
                for (uint32_t g = (o + kRegion - 1) >> kRegionShift; ((uint64_t)g << kRegionShift) < (uint64_t)o + ll; g++) idx[g] = seq;
            }
            r.n++; r.ob += ll; o += ll; seq++;
            p = csize;
            break;
        }
        uint32_t m = q + ll;                     // offset field; m + 2 <= csize - 6
        uint32_t ml = t & 15;
        uint32_t nx = m + 2;
        if (ml == 15) {                          // limit iend-LASTLITERALS+1
            uint32_t b;
            do {
                b = ld_u8(src + nx); nx++; ml += b;
                if (nx + 4 > csize || ml > 0x7FFFFFF0u) { r.err = p + 1; break; }
            } while (b == 255);
            if (r.err) break;
        }
        ml += kMinMatch;
        if (EMIT) {
            const uint32_t off = ld_u16(src + m);
            const uint32_t ms = o + ll;          // match start in the output
            if (off == 0 || off > ms) { r.err = p + 1; break; }          // lz4.c:2356
            if (cap - ms < ml + kLastLiterals) { r.err = p + 1; break; } // lz4.c:2423
            SeqRec rec; rec.outpos = o; rec.litpos = q; rec.ll = ll; rec.off = off;
            tab[seq] = rec;
            for (uint32_t g = (o + kRegion - 1) >> kRegionShift; (g << kRegionShift) < o + ll + ml; g++) idx[g] = seq;
        }
        if (r.ob + ll + ml < r.ob) { r.err = p + 1; break; }             // u32 overflow
        r.n++; r.ob += ll + ml; o += ll + ml; seq++;
        p = nx;
    }
    r.exit = p;
    return r;
}

// ------------------------------------------------------------------------------ phase 5
// 16 bytes as four dwords; byte i of the chunk is byte (i & 3) of dword (i >> 2).
__device__ __forceinline__ uint32_t chunk_byte(const U32x4& a, uint32_t i) {
    const uint32_t d = (i & 8) ? ((i & 4) ? a.w : a.z) : ((i & 4) ? a.y : a.x);
    return (d >> ((i & 3) * 8)) & 0xFFu;
}
__device__ __forceinline__ void chunk_set_byte(U32x4& a, uint32_t i, uint32_t b) {
    const uint32_t sh = (i & 3) * 8, m = ~(0xFFu << sh), v = b << sh;
    switch (i >> 2) {
    case 0: a.x = (a.x & m) | v; break;
    case 1: a.y = (a.y & m) | v; break;
    case 2: a.z = (a.z & m) | v; break;
    default: a.w = (a.w & m) | v; break;
    }
}
// bytes [lo, 16) of the result come from v, bytes [0, lo) from a
__device__ __forceinline__ U32x4 chunk_merge_from(const U32x4& a, const U32x4& v, uint32_t lo) {
    const uint64_t m0 = lo < 8 ? (~0ull << (lo * 8)) : 0ull;
    const uint64_t m1 = lo <= 8 ? ~0ull : (~0ull << ((lo - 8) * 8));
    const uint64_t a0 = (uint64_t)a.x | ((uint64_t)a.y << 32), a1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
    const uint64_t v0 = (uint64_t)v.x | ((uint64_t)v.y << 32), v1 = (uint64_t)v.z | ((uint64_t)v.w << 32);
    const uint64_t r0 = (a0 & ~m0) | (v0 & m0), r1 = (a1 & ~m1) | (v1 & m1);
    U32x4 r; r.x = (uint32_t)r0; r.y = (uint32_t)(r0 >> 32); r.z = (uint32_t)r1; r.w = (uint32_t)(r1 >> 32);
    return r;
}
// 16 bytes of the ring starting at ANY byte position (wraps)
__device__ __forceinline__ U32x4 ring_read16(const uint8_t* ring, uint32_t pos) {
    const uint32_t* r32 = (const uint32_t*)ring;
    const uint32_t b = pos & ~3u, sh = pos & 3u;
    const uint32_t d0 = r32[((b) & kRingMask) >> 2], d1 = r32[((b + 4) & kRingMask) >> 2],
                   d2 = r32[((b + 8) & kRingMask) >> 2], d3 = r32[((b + 12) & kRingMask) >> 2],
                   d4 = r32[((b + 16) & kRingMask) >> 2];
    U32x4 v;
    v.x = align_bytes(d1, d0, sh); v.y = align_bytes(d2, d1, sh);
    v.z = align_bytes(d3, d2, sh); v.w = align_bytes(d4, d3, sh);
    return v;
}

// Is output chunk c (global chunk index = output position / 16) final in the ring?
// g = a lower bound of the first unfinished region (everything below it is final).
__device__ __forceinline__ bool chunk_final(uint32_t c, uint32_t g, const uint32_t* fin, const uint64_t* bits) {
    const uint32_t r = c >> 6;
    if (r < g) return true;
    const uint32_t f = lds_load_acquire(&fin[r & (kDecWaves - 1)]);     // read fin BEFORE bits
    const uint32_t q = r / kDecWaves;
    if (f > q) return true;
    if (f < q) return false;
    const uint64_t b = lds_load_acquire64(&bits[r & (kSlots - 1)]);
    return (b >> (c & 63)) & 1;
}

__device__ __forceinline__ void copy_phase(const uint8_t* __restrict__ src, uint32_t csize, uint8_t* dst,
                                           uint32_t total, const SeqRec* tab, uint32_t nseq,
                                           const uint32_t* idx, char* smem) {
    uint8_t* ring = (uint8_t*)(smem + kOffRing);
    const uint32_t lane = lane_id(), w = wave_id();
    uint32_t* first = (uint32_t*)(smem + kOffFirst) + w * 64;
    uint64_t* bits = (uint64_t*)(smem + kOffBits);
    uint32_t* fin = (uint32_t*)(smem + kOffFin);
    const uint32_t nreg = (uint32_t)(((uint64_t)total + kRegion - 1) >> kRegionShift);
    uint32_t myfin = 0;
    for (uint32_t R = w; R < nreg; R += kDecWaves, myfin++) {
        const uint32_t x0 = R << kRegionShift;
        uint32_t x1 = x0 + kRegion; if (x1 > total || x1 < x0) x1 = total;
        // -- flow control: region R overwrites the ring slot of region R-128, which waves working
        //    on regions <= R-64 may still read.  g = first region not known to be complete.
        uint32_t g;
        for (;;) {
            uint32_t f = lds_load_acquire(&fin[lane & (kDecWaves - 1)]) * kDecWaves + (lane & (kDecWaves - 1));
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)f, d); if (y < f) f = y; }
            g = f;
            if (g + kMaxLead >= R) break;
            spin_pause();
        }
        // -- which sequence covers the first byte of each chunk?  (records j0..jl overlap the region)
        const uint32_t j0 = idx[R];
        const uint32_t jl = (x1 < total) ? idx[R + 1] : nseq - 1;
        first[lane] = 0;
        wave_lds_fence();
        for (uint32_t base = 1; base <= jl - j0; base += 64) {
            const uint32_t r = base + lane;
            if (r <= jl - j0) {
                const uint32_t o = tab[j0 + r].outpos;               // > x0
                const uint32_t s = (o - x0 + kChunk - 1) / kChunk;
                if (s < 64) atomicMax(&first[s], r);
            }
        }
        wave_lds_fence();
        uint32_t j = j0 + wave_incl_max(first[lane]);

        const uint32_t c0 = x0 + kChunk * lane;
        uint32_t c1 = c0 + kChunk; if (c1 > x1) c1 = x1;
        const bool active = c0 < x1;
        const uint32_t mychunk = c0 / kChunk;
        uint32_t pos = c0;
        U32x4 acc; acc.x = acc.y = acc.z = acc.w = 0;
        SeqRec rec, nrec;
        rec.outpos = rec.litpos = rec.ll = rec.off = 0; nrec = rec;
        if (active) { rec = tab[j]; nrec = tab[j + 1]; }
        bool done = !active;
        uint64_t donemask = 0;
        const uint32_t slot = R & (kSlots - 1);
        for (;;) {
            bool newly = false;
            if (!done) {
                while (pos < c1) {
                    if (pos >= nrec.outpos) { j++; rec = nrec; nrec = tab[j + 1]; continue; }
                    const uint32_t lit_end = rec.outpos + rec.ll;
                    const uint32_t lo = pos - c0;
                    if (pos < lit_end) {
                        // ---- literal piece [pos, stop): bytes come from the compressed stream
                        const uint32_t stop = lit_end < c1 ? lit_end : c1;
                        const int32_t A = (int32_t)(rec.litpos + c0 - rec.outpos);   // src index of chunk byte 0
                        if (A >= 0 && (uint32_t)A + 16 <= csize) {
                            acc = chunk_merge_from(acc, ld_global16(src + A), lo);
                        } else {                                     // block edges: byte by byte
                            for (uint32_t i = lo; i < stop - c0; i++) chunk_set_byte(acc, i, ld_u8(src + (A + (int32_t)i)));
                        }
                        pos = stop;
                    } else {
                        // ---- match piece [pos, stop)
                        uint32_t stop = nrec.outpos < c1 ? nrec.outpos : c1;
                        const uint32_t ms = lit_end, off = rec.off, ml = nrec.outpos - ms;
                        if (off >= kChunk) {                       // sources lie in earlier chunks
                            uint32_t s0;                              // source of byte `pos`
                            if (off >= ml) s0 = pos - off;
                            else {                                    // periodic, period >= 16: at most
                                const uint32_t d = (pos - ms) % off;  // one wrap inside a piece
                                s0 = ms - off + d;
                                if (d + (stop - pos) > off) stop = pos + (off - d);
                            }
                            if (!(chunk_final(s0 / kChunk, g, fin, bits) &&
                                  chunk_final((s0 + (stop - pos) - 1) / kChunk, g, fin, bits))) break;
                            acc = chunk_merge_from(acc, ring_read16(ring, s0 - lo), lo);
                        } else {
                            // short offset (< 16): sources lie in [ms-off, ms) (period `off` if overlapping),
                            // possibly inside this very chunk (still in registers)
                            const uint32_t p0 = ms - off;
                            bool ok = true;
                            if (p0 / kChunk < mychunk) ok = chunk_final(p0 / kChunk, g, fin, bits);
                            if (ok && (ms - 1) / kChunk < mychunk && (ms - 1) / kChunk != p0 / kChunk) ok = chunk_final((ms - 1) / kChunk, g, fin, bits);
                            if (!ok) break;
                            uint32_t d = (pos - ms) % off;
                            for (uint32_t i = lo; i < stop - c0; i++) {
                                const uint32_t sp = p0 + d;
                                const uint32_t b = sp >= c0 ? chunk_byte(acc, sp - c0) : (uint32_t)ring[sp & kRingMask];
                                chunk_set_byte(acc, i, b);
                                if (++d == off) d = 0;
                            }
                        }
                        pos = stop;
                    }
                }
                if (pos >= c1) {
                    *(U32x4*)(ring + (c0 & kRingMask)) = acc;
                    if (c1 - c0 == kChunk) st_global16(dst + c0, acc);
                    else for (uint32_t i = 0; i < c1 - c0; i++) dst[c0 + i] = (uint8_t)chunk_byte(acc, i);
                    done = true; newly = true;
                }
            }
            const uint64_t m = __ballot(newly);
            if (m) {
                donemask |= m;
                wave_lds_fence();                       // chunk data before the done bits
                if (lane == 0) lds_store_release64(&bits[slot], donemask);
            }
            if (__all(done)) break;
#ifdef LZ4AMD_TRACE
            { static unsigned long long spins = 0; if (lane == 0 && (++spins % 100000) == 0) fprintf(stderr, "spin R=%u w=%u donemask=%llx g=%u fin0=%u\n", R, w, (unsigned long long)donemask, g, fin[0]); }
            if (!done && R == 0 && lane == 32) { static unsigned long long sp2 = 0; if ((++sp2 % 100000) == 0) fprintf(stderr, "  lane %u pos=%u c0=%u c1=%u j=%u rec(o=%u ll=%u off=%u) next=%u\n", lane, pos, c0, c1, j, rec.outpos, rec.ll, rec.off, nrec.outpos); }
#endif
            spin_pause();
        }
        // region complete: prepare the bits of my next region, then publish
        wave_lds_fence();
        if (lane == 0) {
            lds_store_release64(&bits[(R + kDecWaves) & (kSlots - 1)], 0ull);
            lds_store_release(&fin[w], myfin + 1);
        }
    }
}

__device__ __forceinline__ void decode_one_block(const DecBatch& P, uint32_t b, char* smem) {
    const uint32_t tid = threadIdx.x;
    uint32_t* scan = (uint32_t*)(smem + kOffScan);
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    uint32_t* seg_exit = (uint32_t*)(smem + kOffSegExit);
    uint32_t* seg_n = (uint32_t*)(smem + kOffSegN);
    uint32_t* seg_ob = (uint32_t*)(smem + kOffSegOb);

    const uint8_t* __restrict__ src = P.src[b];
    uint8_t* dst = P.dst[b];
    const int32_t csize_i = P.src_size[b];
    const int32_t cap_i = P.dst_cap[b];

    // -- degenerate inputs (lz4.c:2036, 2062-2069)
    if (src == nullptr || cap_i < 0) { if (tid == 0) P.result[b] = -1; return; }
    if (cap_i == 0) {
        if (tid == 0) P.result[b] = (csize_i == 1 && src[0] == 0) ? 0 : -1;
        return;
    }
    if (csize_i <= 0) { if (tid == 0) P.result[b] = -1; return; }
    const uint32_t csize = (uint32_t)csize_i, cap = (uint32_t)cap_i;

    uint8_t* scratch = P.scratch + (uint64_t)blockIdx.x * P.scratch_stride;
    SeqRec* tab = (SeqRec*)scratch;
    uint32_t* idx = (uint32_t*)(scratch + P.table_bytes);
    uint64_t* prof = P.prof ? P.prof + (uint64_t)blockIdx.x * 8 : nullptr;
    if (prof && tid == 0) prof[0] = clock_ticks();

    // ---------------------------------------------------------------- phase 1: WALK
    uint32_t G = (csize + kDecThreads - 1) / kDecThreads;
    if (G < kMinSeg) G = kMinSeg;
    const uint32_t nseg = (csize + G - 1) / G;
    const bool has_seg = tid < nseg;
    const uint32_t s = tid * G;
    uint32_t e = s + G; if (e > csize || e < s) e = csize;

    uint32_t my_entry = kNone, my_err = 0;
    if (tid == 0) misc[M_ERR] = kNone;
    if (has_seg) {
        WalkOut w; w.exit = 0; w.n = 0; w.ob = 0; w.err = 0;
        if (tid > 0) {
            const uint32_t start = s > kWarm ? s - kWarm : 0;
            w = walk_chain<false>(src, csize, start, s, nullptr, nullptr, 0, 0, 0);
        }
        if (!w.err) {
            my_entry = w.exit;                        // first chain position >= s
            w = walk_chain<false>(src, csize, my_entry, e, nullptr, nullptr, 0, 0, 0);
            my_err = w.err;
            // a malformed chain ends the block: successors just pass through
            seg_exit[tid] = w.err ? csize : w.exit; seg_n[tid] = w.n; seg_ob[tid] = w.ob;
        } else {
            seg_exit[tid] = kNone; seg_n[tid] = 0; seg_ob[tid] = 0;   // unresolved guess
        }
    }
    if (prof && tid == 0) prof[1] = clock_ticks();
    // ---------------------------------------------------------------- phase 2: FIX
    for (;;) {
        __syncthreads();
        uint32_t want = kNone;
        if (has_seg) want = (tid == 0) ? 0u : seg_exit[tid - 1];
        __syncthreads();
        int changed = 0;
        if (has_seg) {
            if (want == kNone) changed = 1;                   // predecessor not resolved yet
            else if (want != my_entry) {
                WalkOut w = walk_chain<false>(src, csize, want, e, nullptr, nullptr, 0, 0, 0);
                my_entry = want; my_err = w.err;
                // a malformed chain stops here; successors then never resolve -> report below
                seg_exit[tid] = w.err ? csize : w.exit; seg_n[tid] = w.n; seg_ob[tid] = w.ob;
                changed = 1;
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
    if (prof && tid == 0) prof[2] = clock_ticks();
    // ---------------------------------------------------------------- phase 3: SCAN
    uint32_t seq0, nseq, out0, total;
    {
        uint64_t out0_64, total_64;
        block_excl_sum2(has_seg ? seg_n[tid] : 0u, has_seg ? (uint64_t)seg_ob[tid] : 0ull, scan,
                        seq0, out0_64, nseq, total_64);
        int bad = 0;
        if (my_err) { atomicMin(&misc[M_ERR], my_err - 1); bad = 1; }
        // output positions beyond the capacity are errors (this also keeps them inside u32)
        if (has_seg && out0_64 + seg_ob[tid] > cap) { atomicMin(&misc[M_ERR], my_entry < csize ? my_entry : csize - 1); bad = 1; }
        out0 = (uint32_t)out0_64; total = (uint32_t)total_64;
        // the vote (not a re-read of misc[M_ERR]) decides, so that phase-4 writers of the next
        // stage cannot race with slow readers of this one
        if (__syncthreads_or(bad)) { if (tid == 0) P.result[b] = err_at(misc[M_ERR]); return; }
    }

    // ---------------------------------------------------------------- phase 4: EMIT
    {
        int bad = 0;
        if (has_seg) {
            WalkOut w = walk_chain<true>(src, csize, my_entry, e, tab, idx, seq0, out0, cap);
            if (w.err) { atomicMin(&misc[M_ERR], w.err - 1); bad = 1; }
        }
        if (tid == 0) { SeqRec rec; rec.outpos = total; rec.litpos = csize; rec.ll = 0; rec.off = 0; tab[nseq] = rec; }
        // phase-5 state (the seg_* arrays that overlay the ring are dead from here on)
        if (tid < kSlots) ((uint64_t*)(smem + kOffBits))[tid] = 0;
        if (tid < kDecWaves) ((uint32_t*)(smem + kOffFin))[tid] = 0;
        // barrier: table + index visible to the whole workgroup
        if (__syncthreads_or(bad)) { if (tid == 0) P.result[b] = err_at(misc[M_ERR]); return; }
    }
    if (tid == 0) P.result[b] = (int32_t)total;
    if (prof && tid == 0) prof[3] = clock_ticks();

    // ---------------------------------------------------------------- phase 5: COPY
    copy_phase(src, csize, dst, total, tab, nseq, idx, smem);
    __syncthreads();                    // the ring is reused by the next block's phase 1
    if (prof && tid == 0) { prof[4] = clock_ticks(); prof[5] = nseq; prof[6] = total; prof[7] = csize; }
}

// Workgroups pull blocks from a device-wide ticket counter (load balance for ragged batches).
__device__ __forceinline__ void decompress_batch_body(const DecBatch& P) {
    LZ4AMD_DYN_LDS(smem);
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) misc[M_BLOCK] = take_ticket(P.ticket);
        __syncthreads();
        const uint32_t b = misc[M_BLOCK];
        if (b >= P.n_blocks) break;
        decode_one_block(P, b, smem);
    }
}

} // namespace lz4amd
