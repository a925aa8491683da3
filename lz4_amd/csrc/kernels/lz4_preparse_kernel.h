// lz4_preparse_kernel.h -- stage A of the LZ4 block decoder (gfx950): the serial token chain of a block becomes a
// table of sequence records.  All 1024 threads of the workgroup take part.
//
// The compressed stream is cut in 1024 SEGMENTS; every thread follows the chain of its own segment, starting 768
// bytes EARLY at an arbitrary byte and relying on LZ4 chains self-synchronising (a wrong start merges with the true
// chain after a few hundred bytes).  A fix-point pass then makes it exact: segment j is right iff it started where
// segment j-1 exited; threads whose guess was wrong re-walk from the true entry (segment 0 starts at byte 0, so by
// induction the result is the true chain for every input; 1-3 rounds on real data).  An accounting walk counts
// sequences and output bytes and applies the input-side format rules (read_variable_length lz4.c:1979-2014, the
// last-literals test lz4.c:2279, 2312-2318), block-wide prefix sums give every segment its first sequence number and
// output position, and a last walk writes the records (output-side rules lz4.c:2356, 2423 applied) - so a malformed
// block is rejected before a byte of output is written.  The warm-up walk runs mostly over literal bytes misread as
// tokens (a step per ~6.5 bytes), so it is a position-only loop with one byte load per trip; rarer token shapes are
// parked and handled every fourth trip.  Blocks whose compressed bytes fit in LDS are walked out of LDS.
#pragma once
#include "lz4_common.h"

namespace lz4amd { namespace pre {

struct alignas(16) SeqRec { uint32_t outpos, litpos, ll, off; };

enum : uint32_t {
    kDecThreads = 1024,
    kSegShift = 8,
    kSeg = 1u << kSegShift,                     // compressed bytes per pre-parse segment (granularity)
    kPreLanes = 1024,                           // pre-parse lanes per block (segments)
    kPreWarm = 768,                             // speculative warm-up distance
    kBias = 65536,                              // output positions are biased: [kBias - prefix, kBias) is the history before dst
    kNone = 0xFFFFFFFFu,
};

// LDS carve-up (bytes)
enum : uint32_t {
    kOffScan = 0,                                            // u32[64] (3 per wave needed)
    kOffMisc = kOffScan + 64 * 4,                            // u32[32]
    kOffSegExit = kOffMisc + 32 * 4,
    kOffRecStage = kOffSegExit + kPreLanes * 4,              // SeqRec[4][kPreLanes]: records wait here to leave four at a time
    kOffCStage = kOffRecStage,                               // the compressed block itself, when it fits (then the records need no staging)
    kPreLdsBytes = 152u << 10,
    kCStageMax = kPreLdsBytes - kOffCStage - 32,             // largest compressed block the pre-parse walks out of LDS
};
static_assert(kOffRecStage + 4 * kPreLanes * 16 <= kPreLdsBytes, "LDS budget");
enum : uint32_t { M_ERR = 1, M_FIRSTBAD = 5 };

// 16 bytes as four dwords; byte i of the chunk is byte (i & 3) of dword (i >> 2).
// (written with selects on whole dwords: indexing the vector dynamically would send it to scratch)
__device__ __forceinline__ uint32_t chunk_byte(const U32x4& a, uint32_t i) {
    const uint32_t lo = (i & 4) ? a[1] : a[0], hi = (i & 4) ? a[3] : a[2];
    const uint32_t d = (i & 8) ? hi : lo;
    return (d >> ((i & 3) * 8)) & 0xFFu;
}
__device__ __forceinline__ void chunk_set_byte(U32x4& a, uint32_t i, uint32_t b) {
    const uint32_t sh = (i & 3) * 8, m = 0xFFu << sh, v = (b & 0xFFu) << sh;
    const uint32_t k = i >> 2;
    a[0] = (k == 0) ? ((a[0] & ~m) | v) : a[0];
    a[1] = (k == 1) ? ((a[1] & ~m) | v) : a[1];
    a[2] = (k == 2) ? ((a[2] & ~m) | v) : a[2];
    a[3] = (k == 3) ? ((a[3] & ~m) | v) : a[3];
}


// scratch of one workgroup: the sequence-record table of the block it is decoding.  Every sequence
// but the last takes >= 3 compressed bytes; +1 last, +1 sentinel.
__host__ __device__ inline uint64_t scratch_bytes(uint32_t max_csize) {
    return ((uint64_t)max_csize / 3 + 4) * sizeof(SeqRec);
}


// The compressed stream as the pre-parse walkers see it: global memory, served by the CU's L1 (few lanes
// walk long stretches - see preparse_block - so their current lines stay L1 resident) - or, for a block
// whose compressed bytes fit beside the pre-parse's own LDS (<= kCStageMax: every block up to 256 KB at
// ratio >= 1.8), a copy of it in LDS: a walk is a chain of dependent loads, ~2 k cycles each from memory,
// ~150 from LDS, and for small blocks that chain (the fixed 768-byte warm-up) is most of the decode time.
struct CView {
    lz4amd_gsrc g;
    const uint8_t* l;           // LDS copy of the block, or nullptr
    uint32_t csize;
    __device__ __forceinline__ uint32_t u8(uint32_t p) const { return l ? (uint32_t)l[p] : (uint32_t)g[p]; }
    __device__ __forceinline__ uint32_t u16(uint32_t p) const { return u8(p) | (u8(p + 1) << 8); }
    __device__ __forceinline__ bool in(uint32_t p) const { return p < csize; }
    __device__ __forceinline__ bool has8(uint32_t p) const { return p < csize && csize - p >= 8; }
    __device__ __forceinline__ uint64_t ld8_if(uint32_t p, bool ok) const {
        uint64_t v = 0;
        if (ok) {
            if (l) {            // three aligned dwords + two v_alignbyte (the copy is padded past csize)
                const uint32_t* w = (const uint32_t*)(l + (p & ~3u));
                const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], sh = p & 3u;
                v = (uint64_t)align_bytes(w1, w0, sh) | ((uint64_t)align_bytes(w2, w1, sh) << 32);
            } else __builtin_memcpy(&v, g + p, 8);
        }
        return v;
    }
};

struct WalkOut { uint32_t exit, n, ob, err; };

// Literal-length field of the token at p (lz4.c:1979-2014, limit iend-15).  q = first literal byte.
// Fast path: token and up to 6 extension bytes in one 8-byte LDS read.
__device__ __forceinline__ bool read_litlen(const CView& V, uint32_t csize, uint32_t p, uint32_t& t,
                                            uint32_t& ll, uint32_t& q) {
    uint64_t w = 0;
    const bool fast = V.has8(p);
    if (fast) { w = V.ld8_if(p, true); t = (uint32_t)w & 0xFFu; } else t = V.u8(p);
    ll = t >> 4; q = p + 1;
    if (ll != 15) return true;
    if (fast) {
        const uint64_t x = w >> 8, inv = ~x & 0x00FFFFFFFFFFFFFFull;     // 7 extension bytes
        const uint32_t k = inv ? ((uint32_t)__ffsll((long long)inv) - 1) >> 3 : 7u;
        if (k < 7) {
            if (q + k + 15 >= csize) return false;                   // the last byte read is q + k
            ll = 15 + 255 * k + ((uint32_t)(x >> (8 * k)) & 0xFFu);
            q += k + 1;
            return true;
        }
    }
    uint32_t b;
    do {
        if (q + 15 >= csize) return false;
        b = V.u8(q); q++; ll += b;
        if (ll > csize) return false;
    } while (b == 255);
    return true;
}
// Offset and match-length field at m (limit iend-LASTLITERALS+1).  nx = next token.
__device__ __forceinline__ bool read_match(const CView& V, uint32_t csize, uint32_t m, uint32_t t,
                                           uint32_t& off, uint32_t& ml, uint32_t& nx) {
    uint64_t y = 0;
    const bool fast = V.has8(m);
    if (fast) { y = V.ld8_if(m, true); off = (uint32_t)y & 0xFFFFu; } else off = V.u16(m);
    ml = t & 15; nx = m + 2;
    if (ml != 15) return true;
    if (fast) {
        const uint64_t z = y >> 16, inv = ~z & 0x0000FFFFFFFFFFFFull;     // 6 extension bytes
        const uint32_t k = inv ? ((uint32_t)__ffsll((long long)inv) - 1) >> 3 : 6u;
        if (k < 6) {
            nx += k + 1;
            if (nx + 4 > csize) return false;
            ml = 15 + 255 * k + ((uint32_t)(z >> (8 * k)) & 0xFFu);
            return true;
        }
    }
    uint32_t b;
    do {
        b = V.u8(nx); nx++; ml += b;
        if (nx + 4 > csize || ml > 0x7FFFFFF0u) return false;
    } while (b == 255);
    return true;
}

// One sequence of the chain, as the walkers need it.
struct SeqStep { uint32_t ll, q, off, ml, nx; bool last, bad; };

// Generic (byte-wise) decode of the sequence at p: any field length, window misses allowed.
__device__ __forceinline__ SeqStep seq_step_slow(const CView& V, uint32_t csize, uint32_t p, uint32_t out_room, bool emit) {
    SeqStep s; s.off = 0; s.ml = 0; s.nx = 0; s.last = false; s.bad = true;
    uint32_t t;
    if (!read_litlen(V, csize, p, t, s.ll, s.q)) return s;
    const uint32_t rem = csize - s.q;
    s.last = (rem < s.ll + 8) || (emit && out_room < s.ll + kMfLimit);
    if (s.last) { s.bad = false; return s; }
    if (!read_match(V, csize, s.q + s.ll, t, s.off, s.ml, s.nx)) return s;
    s.bad = false;
    return s;
}

// Follow the token chain from p while p < e (e <= csize).  err != 0 => malformed at err-1.
// EMIT: also write SeqRec's (ring, from sequence number `seq`, output position `o`) and apply the
// output-side rules (needs cap).  The input-side rules are those of the reference's safe loop
// (lz4.c:1979-2014 length fields, lz4.c:2279 last-literals test).  The common case - both length
// fields and the offset inside two 8-byte LDS reads - is straight-line code with selects; anything
// else (fields longer than 6 extension bytes, bytes outside the LDS window) takes the byte-wise
// path above.
// Records leave through a small LDS staging area, four at a time (64 contiguous bytes per lane): a store
// after every sequence would sit in the same in-order memory queue as the next sequence's loads, and the
// walk would wait for the write latency at every step (measured: 0.8 M of the 1.6 M cycles of this pass).
template <bool EMIT>
__device__ __forceinline__ WalkOut walk_chain(const CView& V, uint32_t csize, uint32_t p, uint32_t e,
                                              SeqRec* recs, uint32_t seq, uint32_t o, uint32_t cap, uint32_t low,
                                              SeqRec* stage = nullptr) {
    WalkOut r; r.n = 0; r.ob = 0; r.err = 0;
    const uint32_t seq0 = seq;
    uint32_t nbuf = 0;
    auto put = [&](const SeqRec& rec) {
        if (stage == nullptr) { recs[seq0 + nbuf] = rec; nbuf++; return; }         // (loads come from LDS: nothing queues behind the store)
        stage[(nbuf & 3) * kPreLanes] = rec;
        nbuf++;
        if ((nbuf & 3) == 0) {
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) recs[seq0 + nbuf - 4 + i] = stage[i * kPreLanes];
        }
    };
    while (p < e) {
        SeqStep s;
        const uint32_t room = EMIT ? cap - o : 0u;
        // ---- token + literal length
        bool slow = !V.has8(p);
        const uint64_t w = V.ld8_if(p, !slow);
        const uint32_t t = (uint32_t)w & 0xFFu, nib = t >> 4;
        const uint64_t x = w >> 8, inv = ~x & 0x00FFFFFFFFFFFFFFull;
        const uint32_t k = inv ? ((uint32_t)__ffsll((long long)inv) - 1) >> 3 : 7u;
        const bool l15 = nib == 15;
        slow = slow || (l15 && k >= 7);
        s.ll = l15 ? 15 + 255 * k + ((uint32_t)(x >> (8 * (k & 7))) & 0xFFu) : nib;
        s.q = p + 1 + (l15 ? k + 1 : 0);
        s.bad = l15 && (p + k + 16 >= csize);                    // extension byte i is read only if q0+i+15 < csize
        const uint32_t rem = csize - s.q;
        s.last = (rem < s.ll + 8) || (EMIT && room < s.ll + kMfLimit);
        // ---- offset + match length
        const uint32_t m = s.q + s.ll;
        const bool need2 = !slow && !s.bad && !s.last;
        const bool ok2 = need2 && V.has8(m);
        slow = slow || (need2 && !ok2);
        const uint64_t y = V.ld8_if(m, ok2);
        s.off = (uint32_t)y & 0xFFFFu;
        const uint64_t z = y >> 16, invz = ~z & 0x0000FFFFFFFFFFFFull;
        const uint32_t km = invz ? ((uint32_t)__ffsll((long long)invz) - 1) >> 3 : 6u;
        const uint32_t mnib = t & 15;
        const bool m15 = mnib == 15;
        slow = slow || (ok2 && m15 && km >= 6);
        s.ml = m15 ? 15 + 255 * km + ((uint32_t)(z >> (8 * (km & 7))) & 0xFFu) : mnib;
        s.nx = m + 2 + (m15 ? km + 1 : 0);
        if (ok2 && m15 && s.nx + 4 > csize) s.bad = true;
        if (slow) s = seq_step_slow(V, csize, p, room, EMIT);
        if (s.bad) { r.err = p + 1; break; }
        if (s.last) {
            if (csize - s.q != s.ll) { r.err = p + 1; break; }    // must end the input exactly
            if (EMIT) {
                if (room < s.ll) { r.err = p + 1; break; }
                SeqRec rec; rec.outpos = o; rec.litpos = s.q; rec.ll = s.ll; rec.off = 0;
                put(rec);
            }
            r.n++; r.ob += s.ll; o += s.ll; seq++;
            p = csize;
            break;
        }
        const uint32_t ml = s.ml + kMinMatch;
        if (EMIT) {
            const uint32_t ms = o + s.ll;        // match start in the output
            if (s.off == 0 || s.off > ms - low) { r.err = p + 1; break; } // lz4.c:2356 (low = first position with history)
            if (cap - ms < ml + kLastLiterals) { r.err = p + 1; break; } // lz4.c:2423
            SeqRec rec; rec.outpos = o; rec.litpos = s.q; rec.ll = s.ll; rec.off = s.off;
            put(rec);
        }
        if (r.ob + s.ll + ml < r.ob) { r.err = p + 1; break; }           // u32 overflow
        r.n++; r.ob += s.ll + ml; o += s.ll + ml; seq++;
        p = s.nx;
    }
    if (EMIT && stage != nullptr) { for (uint32_t i = 0; i < (nbuf & 3); i++) recs[seq0 + (nbuf & ~3u) + i] = stage[i * kPreLanes]; }
    r.exit = p;
    return r;
}

// Position-only generic step: next token position after the sequence at p (csize at the end of
// the block or on any violation - the accounting walk over the true chain reports those).
__device__ __forceinline__ uint32_t next_pos_slow(const CView& V, uint32_t csize, uint32_t p) {
    const SeqStep s = seq_step_slow(V, csize, p, 0, false);
    return (s.bad || s.last) ? csize : s.nx;
}
// Position-only walk from p to the first chain position >= e (e <= csize), at most max_trips loop
// trips (kNone if it did not get there).  This is what the speculative warm-up runs on, mostly over
// literal bytes misread as tokens (one step per ~6.5 bytes), so a trip is a single LDS byte read
// and a handful of VALU instructions: tokens with both nibbles < 15 are stepped over directly;
// anything else (length extensions, window edge, end of block) parks the lane until the next
// multiple-of-4 trip, where all parked lanes take the generic step together.
__device__ __forceinline__ uint32_t walk_pos(const CView& V, uint32_t csize, uint32_t p, uint32_t e, uint32_t max_trips) {
    bool parked = false;
    uint32_t trip = 0;
    while (p < e) {
        if (trip >= max_trips) { p = kNone; break; }
        if (!parked) {
            const bool inwin = V.in(p);
            const uint32_t b = inwin ? V.u8(p) : 0u;
            const uint32_t ll = b >> 4, ml = b & 15;
            if (inwin && ll != 15 && ml != 15 && p + ll + 9 <= csize) p += 3 + ll;
            else parked = true;
        }
        trip++;
        if ((trip & 3) == 0 && parked) { p = next_pos_slow(V, csize, p); parked = false; }
    }
    return p;
}

// The pre-parse walker: positions, sequence count and output bytes only (no offsets, no records).
// Most of its steps are speculative warm-up over literal bytes misread as tokens (about one step
// per 6.5 bytes on datagen data), so a step must be cheap: ONE 8-byte LDS read per loop trip.  A
// lane is either at a token (mode 0: decodes token + literal length, and is done with the sequence
// unless the match length nibble is 15) or at the offset field of a long match (mode 1: decodes the
// match-length extension).  Same input-side rules as walk_chain<false>.
__device__ __forceinline__ WalkOut walk_count(const CView& V, uint32_t csize, uint32_t p, uint32_t e, uint32_t max_trips) {
    WalkOut r; r.n = 0; r.ob = 0; r.err = 0;
    uint32_t rp = p, pend = 0;          // read position; literal length of the sequence in mode 1
    uint32_t trips = 0;
    bool mode1 = false;
    while (p < e) {
        if (trips++ >= max_trips) { p = kNone; break; }           // gave up (unconfirmed re-walk)
        bool slow = !V.has8(rp);
        const uint64_t w = V.ld8_if(rp, !slow);
        uint32_t add_ob = 0, next_p = p, next_rp = rp;
        bool bad = false, last = false, next_mode1 = false, complete = false;
        uint32_t last_ll = 0, last_q = 0;
        if (!mode1) {
            const uint32_t t = (uint32_t)w & 0xFFu, nib = t >> 4;
            const uint64_t x = w >> 8, inv = ~x & 0x00FFFFFFFFFFFFFFull;
            const uint32_t k = inv ? ((uint32_t)__ffsll((long long)inv) - 1) >> 3 : 7u;
            const bool l15 = nib == 15;
            slow = slow || (l15 && k >= 7);
            const uint32_t ll = l15 ? 15 + 255 * k + ((uint32_t)(x >> (8 * (k & 7))) & 0xFFu) : nib;
            const uint32_t q = p + 1 + (l15 ? k + 1 : 0);
            bad = l15 && (p + k + 16 >= csize);
            last = !bad && (csize - q < ll + 8);
            last_ll = ll; last_q = q;
            const uint32_t m = q + ll, mnib = t & 15;
            if (mnib == 15) { next_mode1 = true; next_rp = m; pend = ll; }
            else { complete = true; add_ob = ll + mnib + kMinMatch; next_p = next_rp = m + 2; }
        } else {
            const uint64_t z = w >> 16, invz = ~z & 0x0000FFFFFFFFFFFFull;
            const uint32_t km = invz ? ((uint32_t)__ffsll((long long)invz) - 1) >> 3 : 6u;
            slow = slow || km >= 6;
            const uint32_t nx = rp + 2 + km + 1;
            bad = nx + 4 > csize;
            complete = true;
            add_ob = pend + 15 + 255 * km + ((uint32_t)(z >> (8 * (km & 7))) & 0xFFu) + kMinMatch;
            next_p = next_rp = nx;
        }
        if (slow) {                                     // byte-wise redo of the whole sequence at p
            const SeqStep s = seq_step_slow(V, csize, p, 0, false);
            bad = s.bad; last = !s.bad && s.last; last_ll = s.ll; last_q = s.q;
            complete = true; next_mode1 = false;
            add_ob = s.ll + s.ml + kMinMatch; next_p = next_rp = s.nx;
        }
        if (bad) { r.err = p + 1; break; }
        if (last) {
            if (csize - last_q != last_ll) { r.err = p + 1; break; }
            r.n++; r.ob += last_ll; p = csize;
            break;
        }
        if (complete) {
            if (r.ob + add_ob < r.ob) { r.err = p + 1; break; }          // u32 overflow
            r.n++; r.ob += add_ob;
        }
        p = next_p; rp = next_rp; mode1 = next_mode1;
    }
    r.exit = p;
    return r;
}
// 16 bytes of the compressed stream at position P (tail of the block zero padded)
__device__ __forceinline__ U32x4 load_granule(lz4amd_gsrc src, uint32_t csize, uint32_t P) {
    if (P + 16 <= csize) return ld_global16(src + P);
    U32x4 v; v[0] = v[1] = v[2] = v[3] = 0;
#pragma nounroll
    for (uint32_t i = 0; i < 16 && P + i < csize; i++) chunk_set_byte(v, i, (uint32_t)src[P + i]);
    return v;
}

// ------------------------------------------------------------------------------ stage A
// The whole block at once: kPreLanes lanes, each owning one SEGMENT of G = csize/kPreLanes bytes
// (rounded up to 256).  Long segments are the point: the speculative warm-up is a fixed price per
// lane (~120 slow steps over literals misread as tokens), the true chain inside the segment costs
// one step per ~40 bytes, and with few lanes every lane's current cache line stays in the CU's L1.
// Returns false (uniformly) when the block is malformed; nseq_out / total_out otherwise.
__device__ __forceinline__ bool preparse_block(lz4amd_gsrc src, uint32_t csize, uint32_t cap, uint32_t prefix,
                                               SeqRec* rectab, char* smem,
                                               uint32_t& nseq_out, uint32_t& total_out, uint64_t* prof) {
    uint64_t pt_walk = 0, pt_fix = 0, pt_iters = 0, pt0 = 0;
    const uint32_t tid = threadIdx.x;
    uint32_t* scan = (uint32_t*)(smem + kOffScan);
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    uint32_t* seg_exit = (uint32_t*)(smem + kOffSegExit);
    if (tid == 0) misc[M_ERR] = kNone;
    if (prof) pt0 = clock_ticks();

    uint32_t G = ((csize + kPreLanes - 1) / kPreLanes + kSeg - 1) & ~(kSeg - 1);
    if (G < kSeg) G = kSeg;
    const uint32_t nst = (csize + G - 1) / G;               // <= kPreLanes
    const uint32_t recap = 64 + G / 8;                       // trips an unconfirmed re-walk may take
    CView V; V.g = src; V.csize = csize; V.l = nullptr;
    const bool staged = csize <= kCStageMax;
    if (staged) {
        uint8_t* const cs = (uint8_t*)(smem + kOffCStage);
        for (uint32_t P = 16 * tid; P < csize + 16; P += 16 * kDecThreads) *(U32x4*)(cs + P) = load_granule(src, csize, P);
        V.l = cs;
        __syncthreads();
    }
    const bool has_seg = tid < nst;
    const uint32_t s = tid * G;
    uint32_t e = s + G; if (e > csize || e < s) e = csize;
    // -- 1. positions: speculative entry (warm-up) and exit of every segment
    //    (the segment itself is walked with the accounting walker: sequences, output bytes and format
    //    errors of the LAST walk of a segment are the ones that count, and that walk starts at the true entry)
    uint32_t my_entry = kNone;
    WalkOut w; w.exit = 0; w.n = 0; w.ob = 0; w.err = 0;
    if (has_seg) {
        my_entry = s > 0 ? walk_pos(V, csize, s > kPreWarm ? s - kPreWarm : 0, s, kNone) : 0u;
        w = walk_count(V, csize, my_entry, e, kNone);
        seg_exit[tid] = w.err ? csize : w.exit;               // a malformed chain ends the block
    }
    // -- 2. fix-point: segment j is right iff it started where segment j-1 exited.  F = first
    //    segment that is not; everything before it is the true chain, so X = exit of segment F-1 is
    //    a true chain position: segments the chain jumps over completely (long literal runs) and
    //    the segment X falls into are settled at once; the others re-walk from their predecessor's
    //    current exit, for a bounded number of trips (that exit may still be garbage, and garbage
    //    is slow to walk: they try again once it has settled).  F grows every round.
    bool first_iter = true;
    for (;;) {
        __syncthreads();
        if (prof) { const uint64_t t1 = clock_ticks(); if (first_iter) pt_walk += t1 - pt0; else { pt_fix += t1 - pt0; pt_iters++; } pt0 = t1; first_iter = false; }
        if (tid == 0) misc[M_FIRSTBAD] = nst;
        uint32_t want = kNone;
        if (has_seg) want = (tid == 0) ? 0u : seg_exit[tid - 1];
        __syncthreads();
        if (has_seg && (want != my_entry || want == kNone)) atomicMin(&misc[M_FIRSTBAD], tid);
        __syncthreads();
        const uint32_t F = misc[M_FIRSTBAD];
        if (F >= nst) break;
        const uint32_t X = (F == 0) ? 0u : seg_exit[F - 1];       // != kNone: segment F-1 is right
        __syncthreads();
        if (has_seg && tid >= F) {
            if (X >= e) { my_entry = X; seg_exit[tid] = X; w.n = 0; w.ob = 0; w.err = 0; }   // the chain jumps over this segment
            else if (X >= s) { my_entry = X; w = walk_count(V, csize, X, e, kNone); seg_exit[tid] = w.err ? csize : w.exit; }   // ... enters it at X
            else if (want != kNone && want != my_entry) {
                w = walk_count(V, csize, want, e, recap);
                seg_exit[tid] = w.exit == kNone ? kNone : (w.err ? csize : w.exit);
                my_entry = (w.exit == kNone) ? kNone : want;             // gave up: not resolved yet
            }
        }
    }
    // -- 3. sequence numbers and output positions of the segments
    if (!has_seg) { w.n = 0; w.ob = 0; w.err = 0; }
    uint32_t ea, ta; uint64_t eb, tb;
    block_excl_sum2(w.n, (uint64_t)w.ob, scan, ea, eb, ta, tb);
    int bad = 0;
    if (w.err) { atomicMin(&misc[M_ERR], w.err - 1); bad = 1; }
    // output positions beyond the capacity are errors (this also keeps them inside u32)
    if (has_seg && eb + w.ob > cap) { atomicMin(&misc[M_ERR], my_entry < csize ? my_entry : csize - 1); bad = 1; }
    if (__syncthreads_or(bad)) return false;
    if (prof) { const uint64_t t1 = clock_ticks(); if (tid == 0) prof[5] = t1 - pt0; pt0 = t1; }
    // -- 4. the records, at their final place in the block's table
    if (has_seg && w.n) {
        const WalkOut w2 = walk_chain<true>(V, csize, my_entry, e, rectab, ea, (uint32_t)eb + kBias, cap + kBias, kBias - prefix,
                                            staged ? nullptr : (SeqRec*)(smem + kOffRecStage) + tid);
        if (w2.err) { atomicMin(&misc[M_ERR], w2.err - 1); bad = 1; }
    }
    if (tid == 0) { SeqRec rec; rec.outpos = (uint32_t)tb + kBias; rec.litpos = csize; rec.ll = 0; rec.off = 0; rectab[ta] = rec; }
    if (__syncthreads_or(bad)) return false;
    nseq_out = ta; total_out = (uint32_t)tb;
    if (prof && tid == 0) { prof[6] = pt_walk; prof[7] = pt_fix | (pt_iters << 48); }
    return true;
}

} } // namespace lz4amd::pre
