// lz4_decompress_kernel.h -- batched LZ4 block decompression for gfx950 (MI355X).
//
// Replaces, for a whole batch of independent blocks resident in HBM, what the reference does
// per block in LZ4_decompress_safe (lib/lz4.c:2451 -> LZ4_decompress_generic lz4.c:2023-2445;
// length fields: read_variable_length lz4.c:1979-2014; end-of-block rules lz4.c:2276-2330,
// 2421-2429).  Accepts ANY legal LZ4 block (reference-produced included), rejects what the
// reference's safe loop rejects, never reads outside src[0,csize) nor writes outside
// dst[0,cap).  This is not a port: the reference decoder is one serial token chain per block;
// here ONE 1024-thread workgroup decodes a block in five data-parallel phases:
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
//             into an L2-resident table, applying the output-side format rules.
//   5 COPY    output-stationary: the output is produced in tiles of <=32 KB through a 128 KB
//             LDS ring that always holds the 64 KB LZ4 window; thread k owns 32 consecutive
//             output bytes, copies its literal bytes straight from the compressed stream, then
//             resolves its match bytes from the ring as soon as the 32-byte regions they read
//             are complete (per-region done bits in LDS, release/acquire, no barriers inside a
//             tile: pure dataflow, dependencies always point backwards so it cannot deadlock).
//             Finished tiles leave for HBM as coalesced 16-byte stores.
//
// HBM traffic per block: compressed bytes read ~2x (walk reads tokens only; copy reads the
// literals), output written once, matches never touch HBM.  No MFMA: this is byte shuffling.
#pragma once
#include "lz4_common.h"
#include "../lz4amd_params.h"

namespace lz4amd {

using DecBatch = ::lz4amd_dec_params;     // argument block (lz4amd_params.h)

struct alignas(16) SeqRec { uint32_t outpos, litpos, ll, off; };

enum : uint32_t {
    kDecThreads = 1024,
    kRingBytes = 128u << 10,
    kRingMask = kRingBytes - 1,
    kRegion = 32,                               // output bytes owned by one thread per tile
    kTileBytes = kDecThreads * kRegion,         // 32 KB
    kSlice = kDecThreads,                       // sequence records staged per tile
    kWarm = 2048,                               // speculative warm-up distance
    kMinSeg = 2048,                             // minimum segment length
    kNone = 0xFFFFFFFFu,
};

// LDS carve-up (bytes)
enum : uint32_t {
    kOffRing = 0,
    kOffSlice = kOffRing + kRingBytes,                       // SeqRec[kSlice + 1]
    kOffRf = kOffSlice + (kSlice + 2) * 16,                  // u32[kDecThreads]
    kOffFlags = kOffRf + kDecThreads * 4,                    // u32[kDecThreads/32]
    kOffScan = kOffFlags + (kDecThreads / 32) * 4,           // u32[64] (3 per wave needed)
    kOffMisc = kOffScan + 64 * 4,                            // u32[16]
    kDecLdsBytes = kOffMisc + 16 * 4,
    // phase 1-4 arrays overlay the ring (not live at the same time)
    kOffSegEntry = 0,
    kOffSegExit = kOffSegEntry + kDecThreads * 4,
    kOffSegN = kOffSegExit + kDecThreads * 4,
    kOffSegOb = kOffSegN + kDecThreads * 4,
};
enum : uint32_t { M_BLOCK = 0, M_ERR = 1 };

__host__ __device__ inline uint64_t dec_scratch_bytes(uint32_t max_csize) {
    // every sequence but the last takes >= 3 compressed bytes; +1 last, +1 sentinel
    return ((uint64_t)max_csize / 3 + 4) * sizeof(SeqRec);
}

struct WalkOut { uint32_t exit, n, ob, err; };

// Follow the token chain from p while p < e (e <= csize).  err != 0 => malformed at err-1.
// EMIT: also write SeqRec's from index `seq` / output position `o`, and apply the output-side
// rules (needs cap).  The input-side rules are those of the reference's safe loop.
template <bool EMIT>
__device__ __forceinline__ WalkOut walk_chain(const uint8_t* __restrict__ src, uint32_t csize,
                                              uint32_t p, uint32_t e, SeqRec* __restrict__ tab,
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
        }
        if (r.ob + ll + ml < r.ob) { r.err = p + 1; break; }             // u32 overflow
        r.n++; r.ob += ll + ml; o += ll + ml; seq++;
        p = nx;
    }
    r.exit = p;
    return r;
}

__device__ __forceinline__ void decode_one_block(const DecBatch& P, uint32_t b, char* smem) {
    const uint32_t tid = threadIdx.x;
    uint8_t* ring = (uint8_t*)(smem + kOffRing);
    SeqRec* slice = (SeqRec*)(smem + kOffSlice);
    uint32_t* rf = (uint32_t*)(smem + kOffRf);
    uint32_t* flags = (uint32_t*)(smem + kOffFlags);
    uint32_t* scan = (uint32_t*)(smem + kOffScan);
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    uint32_t* seg_entry = (uint32_t*)(smem + kOffSegEntry);
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

    SeqRec* tab = (SeqRec*)(P.scratch + (uint64_t)blockIdx.x * P.scratch_stride);

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
            w = walk_chain<false>(src, csize, start, s, nullptr, 0, 0, 0);
        }
        if (!w.err) {
            my_entry = w.exit;                        // first chain position >= s
            w = walk_chain<false>(src, csize, my_entry, e, nullptr, 0, 0, 0);
            my_err = w.err;
            // a malformed chain ends the block: successors just pass through
            seg_exit[tid] = w.err ? csize : w.exit; seg_n[tid] = w.n; seg_ob[tid] = w.ob;
        } else {
            seg_exit[tid] = kNone; seg_n[tid] = 0; seg_ob[tid] = 0;   // unresolved guess
        }
    }
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
                WalkOut w = walk_chain<false>(src, csize, want, e, nullptr, 0, 0, 0);
                my_entry = want; my_err = w.err;
                // a malformed chain stops here; successors then never resolve -> report below
                seg_exit[tid] = w.err ? csize : w.exit; seg_n[tid] = w.n; seg_ob[tid] = w.ob;
                changed = 1;
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
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
            WalkOut w = walk_chain<true>(src, csize, my_entry, e, tab, seq0, out0, cap);
            if (w.err) { atomicMin(&misc[M_ERR], w.err - 1); bad = 1; }
        }
        if (tid == 0) { SeqRec rec; rec.outpos = total; rec.litpos = csize; rec.ll = 0; rec.off = 0; tab[nseq] = rec; }
        // barrier: table visible to the whole workgroup
        if (__syncthreads_or(bad)) { if (tid == 0) P.result[b] = err_at(misc[M_ERR]); return; }
    }
    if (tid == 0) P.result[b] = (int32_t)total;

    // ---------------------------------------------------------------- phase 5: COPY
    const bool dst_aligned = (((uintptr_t)dst) & 15) == 0;
    uint32_t i0 = 0, tb = 0;
    while (tb < total) {
        // -- stage up to kSlice records starting at the sequence that contains byte tb
        const uint32_t idx = i0 + tid;
        if (idx <= nseq) slice[tid] = tab[idx];
        if (tid == 0) { uint32_t j = i0 + kSlice; if (j > nseq) j = nseq; slice[kSlice] = tab[j]; }
        if (tid < kDecThreads / 32) flags[tid] = 0;
        rf[tid] = 0;
        uint32_t lim = tb + kTileBytes; if (lim > total || lim < tb) lim = total;
        const int in_tile = (idx < nseq) && (slice[tid].outpos < lim);
        const uint32_t m = (uint32_t)__syncthreads_count(in_tile);    // >= 1
        uint32_t te = lim;
        if (m == kSlice && slice[kSlice].outpos < te) te = slice[kSlice].outpos;
        const uint32_t i_next = (slice[m].outpos == te) ? i0 + m : i0 + m - 1;

        // -- which sequence covers the first byte of each 32-byte region?
        if (tid > 0 && tid < m) {
            const uint32_t o = slice[tid].outpos;
            const uint32_t r = (o - tb + kRegion - 1) / kRegion;
            if (r < kDecThreads && tb + r * kRegion < slice[tid + 1].outpos) atomicMax(&rf[r], tid);
        }
        __syncthreads();
        const uint32_t first = block_incl_max(rf[tid], scan);

        const uint32_t rb = tb + tid * kRegion;
        uint32_t re = rb + kRegion; if (re > te) re = te;
        const bool active = rb < te;

        // -- pass A: literal bytes come straight from the compressed stream
        if (active) {
            uint32_t j = first, pos = rb;
            while (pos < re) {
                const SeqRec rec = slice[j];
                const uint32_t nexto = slice[j + 1].outpos;
                if (pos >= nexto) { j++; continue; }
                const uint32_t lit_end = rec.outpos + rec.ll;
                if (pos < lit_end) {
                    const uint32_t stop = lit_end < re ? lit_end : re;
                    const uint8_t* sp = src + rec.litpos + (pos - rec.outpos);
                    for (uint32_t k = 0; k < stop - pos; k++) ring[(pos + k) & kRingMask] = sp[k];
                    pos = stop;
                } else {
                    pos = nexto < re ? nexto : re;      // match bytes: pass B
                }
            }
        }
        // -- pass B: match bytes, dataflow over per-region done bits
        {
            uint32_t j = first, pos = rb;
            bool done = !active;
            uint32_t idle = 0;                          // safety net: never spin forever
            for (;;) {
                if (!done) {
                    bool blocked = false;
                    while (pos < re && !blocked) {
                        const SeqRec rec = slice[j];
                        const uint32_t nexto = slice[j + 1].outpos;
                        if (pos >= nexto) { j++; continue; }
                        const uint32_t ms = rec.outpos + rec.ll;       // match start
                        if (pos < ms) { pos = ms < re ? ms : re; continue; }
                        const uint32_t stop = nexto < re ? nexto : re;
                        const uint32_t n = stop - pos;
                        const uint32_t off = rec.off;
                        const uint32_t ml = nexto - ms;
                        if (off >= ml) {
                            // plain copy: sources [pos-off, pos-off+n) lie strictly below pos
                            const uint32_t s0 = pos - off, s1 = s0 + n;
                            if (s1 > tb) {
                                const uint32_t lo = s0 > tb ? s0 : tb;
                                const uint32_t r0 = (lo - tb) / kRegion, r1 = (s1 - 1 - tb) / kRegion;
                                for (uint32_t r = r0; r <= r1; r++)
                                    if (r != tid && !((lds_load_acquire(&flags[r >> 5]) >> (r & 31)) & 1)) blocked = true;
                            }
                            if (!blocked) {
                                for (uint32_t k = 0; k < n; k++)
                                    ring[(pos + k) & kRingMask] = ring[(pos + k - off) & kRingMask];
                                pos = stop;
                            }
                        } else {
                            // overlapping match = periodic replication of the `off` bytes before ms
                            uint32_t d = (pos - ms) % off;
                            uint32_t dd = d;
                            for (uint32_t k = 0; k < n; k++) {
                                const uint32_t sp = ms - off + dd;
                                if (sp >= tb) {
                                    const uint32_t r = (sp - tb) / kRegion;
                                    if (r != tid && !((lds_load_acquire(&flags[r >> 5]) >> (r & 31)) & 1)) blocked = true;
                                }
                                if (++dd == off) dd = 0;
                            }
                            if (!blocked) {
                                for (uint32_t k = 0; k < n; k++) {
                                    ring[(pos + k) & kRingMask] = ring[(ms - off + d) & kRingMask];
                                    if (++d == off) d = 0;
                                }
                                pos = stop;
                            }
                        }
                    }
                    if (pos >= re) {
                        done = true;
                        lds_or_release(&flags[tid >> 5], 1u << (tid & 31));
                    }
                }
                if (__all(done)) break;
                if (++idle > (1u << 22)) { misc[M_ERR] = 0x7FFFFFF0u; break; }   // cannot happen: see header
                spin_pause();
            }
        }
        __syncthreads();
        // -- flush the finished tile: coalesced 16-byte stores
        for (uint32_t x = (tb & ~15u) + 16 * tid; x < te; x += 16 * kDecThreads) {
            if (x >= tb && x + 16 <= te && dst_aligned) {
                *(U32x4*)(dst + x) = *(const U32x4*)(ring + (x & kRingMask));
            } else {
                const uint32_t a = x > tb ? x : tb, z = x + 16 < te ? x + 16 : te;
                for (uint32_t y = a; y < z; y++) dst[y] = ring[y & kRingMask];
            }
        }
        tb = te; i0 = i_next;
        __syncthreads();
        if (misc[M_ERR] != kNone) { if (tid == 0) P.result[b] = err_at(misc[M_ERR]); return; }
    }
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
