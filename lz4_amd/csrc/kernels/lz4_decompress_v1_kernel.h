// lz4_decompress_v1_kernel.h -- ROUND-1 DECODER, kept only for A/B timing against the streaming decoder
// (lz4_decompress_kernel.h); selected with LZ4AMD_DEC=v1.  Batched LZ4 block decompression for gfx950 (MI355X).
//
// Replaces, for a whole batch of independent blocks resident in HBM, what the reference does
// per block in LZ4_decompress_safe (lib/lz4.c:2451 -> LZ4_decompress_generic lz4.c:2023-2445;
// length fields: read_variable_length lz4.c:1979-2014; end-of-block rules lz4.c:2276-2330,
// 2421-2429).  Accepts ANY legal LZ4 block (reference-produced included), rejects what the
// reference's safe loop rejects, never reads outside src[0,csize) nor writes outside
// dst[0,cap).  This is not a port: the reference decoder is one serial token chain per block.
// Here ONE 1024-thread workgroup (16 waves, one CU, ~155 KB of its LDS) decodes a block in two
// stages:
//
//   A PRE-PARSE  turns the serial token chain into a table of sequence records (output position,
//       literal source, literal length, offset; 16 B each) in the workgroup's scratch (L2/HBM).
//       The compressed stream is cut in 1024 SEGMENTS; every thread follows the chain of its own
//       segment, starting 768 bytes EARLY at an arbitrary byte and relying on LZ4 chains
//       self-synchronising (a wrong start merges with the true chain after a few hundred bytes).
//       A fix-point pass then makes it exact: segment j is right iff it started where segment
//       j-1 exited; threads whose guess was wrong re-walk from the true entry (segment 0 starts at
//       byte 0, so by induction the result is the true chain for every input; 1-3 rounds on real
//       data).  An accounting walk counts sequences and output bytes and applies the input-side
//       format rules, block-wide prefix sums give every segment its first sequence number and
//       output position, and a last walk writes the records (output-side rules applied) - so a
//       malformed block is rejected before a byte of output is written.  The warm-up walk runs
//       mostly over literal bytes misread as tokens (a step per ~6.5 bytes), so it is a
//       position-only loop with one byte load per trip; rarer token shapes are parked and handled
//       every fourth trip.
//
//   B STREAM     iterates over the block; every iteration
//       LOAD   refills a 32 KB window of the compressed stream and up to 1023 rows of a record
//              ring in LDS (the global loads are issued before the previous iteration's copy and
//              committed after it),
//       INDEX  notes, for every 1 KB REGION of output the records cover, the record that holds
//              its first byte,
//       COPY   is output-stationary and barrier-free: wave w owns regions w, w+16, ... and lane
//              k owns the 16-byte CHUNK k of the region.  A lane composes its chunk in four
//              VGPRs: literal pieces are unaligned 16-byte reads from the compressed window in
//              LDS (HBM only for literal runs too long to be resident), match pieces are
//              unaligned 16-byte reads from a 96 KB LDS ring that always holds the 64 KB LZ4
//              window; finished chunks go to the ring and to HBM (one 16-byte store per lane,
//              1 KB contiguous per wave).  Match sources that are not final yet are waited for
//              through per-chunk done bits and per-wave completed-region counters in LDS -
//              pure dataflow between the 16 waves.  Dependencies always point to lower output
//              positions and every wave walks its regions in increasing order, so the lowest
//              unfinished region can always finish: no deadlock.
//
// HBM/L2 traffic per block: compressed bytes read ~2.4x (pre-parse walks, stream window), the
// record table written and read once (16 B per sequence), output written once; matches and
// (except for giant runs) literals never touch HBM during the copy.  No MFMA: byte shuffling.
#pragma once
#include "lz4_common.h"
#include "../lz4amd_params.h"

namespace lz4amd { namespace v1 {

using DecBatch = ::lz4amd_dec_params;     // argument block (lz4amd_params.h)

struct alignas(16) SeqRec { uint32_t outpos, litpos, ll, off; };

enum : uint32_t {
    kDecThreads = 1024,
    kDecWaves = kDecThreads / 64,
    kSegShift = 8,
    kSeg = 1u << kSegShift,                     // compressed bytes per pre-parse segment
    kChunk = 16,                                // output bytes owned by one lane per region
    kRegionShift = 10,
    kRegion = 1u << kRegionShift,               // 64 lanes x 16 bytes
    kSlots = 96,                                // output ring slots (regions)
    kRingBytes = kSlots * kRegion,
    kRingPad = 32,                              // mirror of the first bytes: reads never wrap
    kMaxLead = 31,                              // a wave may lead the slowest by this many regions
    kCrBytes = 32u << 10,                       // compressed window (direct mapped by position)
    kCrMask = kCrBytes - 1,
    kRecCap = 1024,                             // sequence-record ring
    kRecMask = kRecCap - 1,
    kIdxCap = 256,                              // regions handled per stream iteration (max)
    kIdxRing = 512,
    kIdxMask = kIdxRing - 1,
    kPreLanes = 1024,                           // pre-parse lanes per block (segments)
    kPreWarm = 768,                             // speculative warm-up distance
    kBias = 65536,                              // output positions are biased: [kBias - prefix, kBias) is the history before dst
    kNone = 0xFFFFFFFFu,
};

// LDS carve-up (bytes)
enum : uint32_t {
    kOffScan = 0,                                            // u32[64] (3 per wave needed)
    kOffMisc = kOffScan + 64 * 4,                            // u32[32]
    kOffPhase = kOffMisc + 32 * 4,
    // stage B
    kOffRing = kOffPhase,
    kOffCr = kOffRing + kRingBytes + kRingPad,
    kOffRecs = kOffCr + kCrBytes,
    kOffIdx = kOffRecs + kRecCap * 16,                       // u32[kIdxRing]
    kOffFirst = kOffIdx + kIdxRing * 4,                      // u32[kDecWaves][64]
    kOffBits = kOffFirst + kDecWaves * 64 * 4,               // DoneEnt[kSlots] per-chunk done bits
    kOffFin = kOffBits + kSlots * 16,                        // u32[kDecWaves] regions completed
    kStreamEnd = kOffFin + kDecWaves * 4,
    // stage A (overlays stage B's arrays; not live at the same time)
    kOffSegExit = kOffPhase,
    kOffRecStage = kOffSegExit + kPreLanes * 4,              // SeqRec[4][kPreLanes]: records wait here to leave four at a time
    kPreEnd = kOffRecStage + 4 * kPreLanes * 16,
    kOffCStage = kOffRecStage,                               // the compressed block itself, when it fits (then the records need no staging)
    kDecLdsBytes = kStreamEnd > kPreEnd ? kStreamEnd : kPreEnd,
    kCStageMax = kDecLdsBytes - kOffCStage - 32,             // largest compressed block the pre-parse walks out of LDS
};
enum : uint32_t { M_BLOCK = 0, M_ERR = 1, M_CARRY = 2, M_HEAD = 3, M_EMIT = 4, M_FIRSTBAD = 5 };   // M_CARRY unused

// scratch of one workgroup: the sequence-record table of the block it is decoding.  Every sequence
// but the last takes >= 3 compressed bytes; +1 last, +1 sentinel.
__host__ __device__ inline uint64_t dec_scratch_bytes(uint32_t max_csize) {
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

__device__ __forceinline__ void chunk_set_byte(U32x4& a, uint32_t i, uint32_t b);
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

// ------------------------------------------------------------------------------ stage B: COPY
// 16 bytes as four dwords; byte i of the chunk is byte (i & 3) of dword (i >> 2).
// (written with selects on whole dwords: indexing the struct dynamically would send it to scratch)
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
// bytes [lo, 16) of the result come from v, bytes [0, lo) from a
__device__ __forceinline__ U32x4 chunk_merge_from(const U32x4& a, const U32x4& v, uint32_t lo) {
    const uint64_t m0 = lo < 8 ? (~0ull << (lo * 8)) : 0ull;
    const uint64_t m1 = lo <= 8 ? ~0ull : (~0ull << ((lo - 8) * 8));
    const uint64_t a0 = (uint64_t)a[0] | ((uint64_t)a[1] << 32), a1 = (uint64_t)a[2] | ((uint64_t)a[3] << 32);
    const uint64_t v0 = (uint64_t)v[0] | ((uint64_t)v[1] << 32), v1 = (uint64_t)v[2] | ((uint64_t)v[3] << 32);
    const uint64_t r0 = (a0 & ~m0) | (v0 & m0), r1 = (a1 & ~m1) | (v1 & m1);
    U32x4 r; r[0] = (uint32_t)r0; r[1] = (uint32_t)(r0 >> 32); r[2] = (uint32_t)r1; r[3] = (uint32_t)(r1 >> 32);
    return r;
}
// 16 bytes starting at ANY byte of an LDS array of dwords (the caller guarantees b+20 in range)
__device__ __forceinline__ U32x4 lds_read16_at(const uint8_t* base, uint32_t a) {
    const uint32_t* r32 = (const uint32_t*)(base + (a & ~3u));
    const uint32_t sh = a & 3u;
    const uint32_t d0 = r32[0], d1 = r32[1], d2 = r32[2], d3 = r32[3], d4 = r32[4];
    U32x4 v;
    v[0] = align_bytes(d1, d0, sh); v[1] = align_bytes(d2, d1, sh);
    v[2] = align_bytes(d3, d2, sh); v[3] = align_bytes(d4, d3, sh);
    return v;
}
// ring address of output position pos
__device__ __forceinline__ uint32_t ring_addr(uint32_t pos) {
    return (((pos >> kRegionShift) % kSlots) << kRegionShift) | (pos & (kRegion - 1));
}
// 16 ring bytes of which byte `lo` is output position s0 (bytes below lo are don't-care and may
// lie before the ring's start: step back with wrap-around; the pad mirrors the ring's first bytes)
__device__ __forceinline__ U32x4 ring_read16(const uint8_t* ring, uint32_t s0, uint32_t lo) {
    uint32_t a = ring_addr(s0);
    a = a >= lo ? a - lo : a + kRingBytes - lo;
    return lds_read16_at(ring, a);
}

// Done tracking: one 16-byte entry per ring slot = {u64 mask of finished chunks, u32 tag, pad}.
// tag = region number + kSlots of the region the mask belongs to (so a slot that was never used
// carries tag = slot index = "region slot-kSlots").  Single writer (lane 0 of the owning wave);
// readers take the entry with one 16-byte LDS read.
struct alignas(16) DoneEnt { uint64_t mask; uint32_t tag, pad; };

// Is output chunk c (global chunk index = output position / 16) final in the ring?
// g = a lower bound of the first unfinished region (everything below it is final).
__device__ __forceinline__ bool chunk_final(uint32_t c, uint32_t g, const DoneEnt* ents) {
    const uint32_t r = c >> 6;
    if (r < g) return true;
    const DoneEnt e = lds_load_ent(&ents[r % kSlots]);
    const uint32_t want = r + kSlots;
    // tag > want: the slot already serves a later region, so r was finished long ago
    return e.tag > want || (e.tag == want && ((e.mask >> (c & 63)) & 1));
}
// both chunks ca <= cb (cb - ca <= 1)
__device__ __forceinline__ bool chunks_final(uint32_t ca, uint32_t cb, uint32_t g, const DoneEnt* ents) {
    if ((cb >> 6) < g) return true;
    if ((ca >> 6) == (cb >> 6)) {
        const uint32_t r = ca >> 6;
        const DoneEnt e = lds_load_ent(&ents[r % kSlots]);
        const uint32_t want = r + kSlots;
        const uint64_t need = (1ull << (ca & 63)) | (1ull << (cb & 63));
        return e.tag > want || (e.tag == want && (e.mask & need) == need);
    }
    return chunk_final(ca, g, ents) && chunk_final(cb, g, ents);
}

struct CopyCtx {
    lz4amd_gsrc src; uint32_t csize; lz4amd_gdst dst;
    uint32_t out_emit;          // output position covered by the records in the ring
    uint32_t rec_head;          // number of records emitted so far (the ring holds a sentinel there)
    uint32_t cr_lo, cr_hi;      // compressed bytes resident in the LDS window
    uint32_t r_ready;           // regions below this one can be composed in this iteration
};

// One wave composes its regions R, R+16, ... below C.r_ready.  R / myfin persist across iterations.
__device__ __forceinline__ void copy_regions(const CopyCtx& C, char* smem, uint32_t& R, uint32_t& myfin) {
    uint8_t* ring = (uint8_t*)(smem + kOffRing);
    const SeqRec* tab = (const SeqRec*)(smem + kOffRecs);
    const uint32_t* idx = (const uint32_t*)(smem + kOffIdx);
    const uint32_t lane = lane_id(), w = wave_id();
    uint32_t* first = (uint32_t*)(smem + kOffFirst) + w * 64;
    DoneEnt* ents = (DoneEnt*)(smem + kOffBits);
    uint32_t* fin = (uint32_t*)(smem + kOffFin);
    const lz4amd_gsrc src = C.src;
    const lz4amd_gdst dst = C.dst;
    const uint32_t csize = C.csize, cr_lo = C.cr_lo, cr_hi = C.cr_hi, cr_span = C.cr_hi - C.cr_lo;
    for (; R < C.r_ready; R += kDecWaves, myfin++) {
        const uint32_t x0 = R << kRegionShift;
        uint32_t x1 = x0 + kRegion; if (x1 > C.out_emit) x1 = C.out_emit;     // only the block's last region is short
        // -- flow control: region R overwrites the ring slot of region R-96, which waves working
        //    on regions <= R-32 may still read.  g = first region not known to be complete.
        uint32_t g;
        for (;;) {
            uint32_t f = lds_load_acquire(&fin[lane & (kDecWaves - 1)]) * kDecWaves + (lane & (kDecWaves - 1));
            g = __builtin_amdgcn_readfirstlane(row16_min_u32(f));   // one value for the whole wave
            if (g + kMaxLead >= R) break;
            spin_pause();
        }
        const uint32_t slot = R % kSlots;
        // the slot is mine now: no chunk of region R is done (mask first, then the tag)
        if (lane == 0) { lds_store_release64(&ents[slot].mask, 0ull); lds_store_release(&ents[slot].tag, R + kSlots); }
        // -- which sequence covers the first byte of each chunk?  (records j0..jl overlap the region)
        const uint32_t j0 = idx[R & kIdxMask];
        const uint32_t jl = (x1 < C.out_emit) ? idx[(R + 1) & kIdxMask] : C.rec_head - 1;
        first[lane] = 0;
        wave_lds_fence();
        for (uint32_t base = 1; base <= jl - j0; base += 64) {
            const uint32_t r = base + lane;
            if (r <= jl - j0) {
                const uint32_t o = tab[(j0 + r) & kRecMask].outpos;  // > x0
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
        U32x4 acc; acc[0] = acc[1] = acc[2] = acc[3] = 0;
        SeqRec rec, nrec;
        rec.outpos = rec.litpos = rec.ll = rec.off = 0; nrec = rec;
        if (active) { rec = tab[j & kRecMask]; nrec = tab[(j + 1) & kRecMask]; }
        bool done = !active;
        uint64_t donemask = __ballot(!active);
        const uint32_t my_ring = (slot << kRegionShift) + kChunk * lane;
        for (;;) {
            bool newly = false;
            if (!done) {
                while (pos < c1) {
                    // one PIECE per trip: the part of a literal run or of a match inside my chunk
                    const uint32_t lit_end = rec.outpos + rec.ll;
                    const uint32_t lo = pos - c0;
                    const bool is_lit = pos < lit_end;
                    const uint32_t pend = is_lit ? lit_end : nrec.outpos;
                    uint32_t stop = pend < c1 ? pend : c1;
                    // kind 0: 16 bytes from LDS at `a` (literals in the window / final long match);
                    // 1: wait; 2: literals from HBM; 3: short or overlapping match
                    uint32_t kind, a = 0;
                    const uint32_t A = rec.litpos + c0 - rec.outpos;     // compressed index of chunk byte 0 (may wrap below 0)
                    const uint32_t off = rec.off, ms = lit_end;
                    if (is_lit) {
                        const bool inwin = (A - cr_lo < cr_span) && (A + 16 <= cr_hi) && ((A & kCrMask) + 20 <= kCrBytes);
                        kind = inwin ? 0u : 2u;
                        a = kOffCr + (A & kCrMask);
                    } else if (off >= kChunk && off >= nrec.outpos - ms) {
                        const uint32_t s0 = pos - off;
                        kind = chunks_final(s0 / kChunk, (s0 + (stop - pos) - 1) / kChunk, g, ents) ? 0u : 1u;
                        uint32_t ra = ring_addr(s0);
                        ra = ra >= lo ? ra - lo : ra + kRingBytes - lo;
                        a = kOffRing + ra;
                    } else kind = 3;
                    if (kind == 0) {
                        acc = chunk_merge_from(acc, lds_read16_at((const uint8_t*)smem, a), lo);
                    } else if (kind == 1) {
                        break;
                    } else if (kind == 2) {
                        if (A < csize && A + 16 <= csize) acc = chunk_merge_from(acc, ld_global16(src + A), lo);
                        else {                                       // block edges: byte by byte
#pragma nounroll
                            for (uint32_t i = lo; i < stop - c0; i++) chunk_set_byte(acc, i, (uint32_t)src[A + i]);
                        }
                    } else {
                        const uint32_t ml = nrec.outpos - ms;
                        if (off >= kChunk) {                       // overlapping, period >= 16: at most one wrap inside a piece
                            const uint32_t d = (pos - ms) % off;
                            const uint32_t s0 = ms - off + d;
                            if (d + (stop - pos) > off) stop = pos + (off - d);
                            if (!chunks_final(s0 / kChunk, (s0 + (stop - pos) - 1) / kChunk, g, ents)) break;
                            acc = chunk_merge_from(acc, ring_read16(ring, s0, lo), lo);
                        } else {
                            // short offset (< 16): sources lie in [ms-off, ms) (period `off` if overlapping),
                            // possibly inside this very chunk (still in registers)
                            (void)ml;
                            const uint32_t p0 = ms - off;
                            bool ok = true;
                            if (p0 / kChunk < mychunk) ok = chunk_final(p0 / kChunk, g, ents);
                            if (ok && (ms - 1) / kChunk < mychunk && (ms - 1) / kChunk != p0 / kChunk) ok = chunk_final((ms - 1) / kChunk, g, ents);
                            if (!ok) break;
                            uint32_t d = (pos - ms) % off;
#pragma nounroll
                            for (uint32_t i = lo; i < stop - c0; i++) {
                                const uint32_t sp = p0 + d;
                                const uint32_t b = sp >= c0 ? chunk_byte(acc, sp - c0) : (uint32_t)ring[ring_addr(sp)];
                                chunk_set_byte(acc, i, b);
                                if (++d == off) d = 0;
                            }
                        }
                    }
                    pos = stop;
                    if (pos < c1 && pos >= nrec.outpos) { j++; rec = nrec; nrec = tab[(j + 1) & kRecMask]; }
                }
                if (pos >= c1) {
                    *(U32x4*)(ring + my_ring) = acc;
                    if (my_ring < kRingPad) *(U32x4*)(ring + kRingBytes + my_ring) = acc;   // mirror
                    if (c1 - c0 == kChunk) st_global16(dst + (c0 - kBias), acc);
                    else {
#pragma nounroll
                        for (uint32_t i = 0; i < c1 - c0; i++) dst[c0 - kBias + i] = (uint8_t)chunk_byte(acc, i);
                    }
                    done = true; newly = true;
                }
            }
            const uint64_t m = __ballot(newly);
            if (m) {
                donemask |= m;
                wave_lds_fence();                       // chunk data before the done bits
                if (lane == 0) lds_store_release64(&ents[slot].mask, donemask);
            }
            if (__all(done)) break;
            spin_pause();
        }
        // region complete
        wave_lds_fence();
        if (lane == 0) lds_store_release(&fin[w], myfin + 1);
    }
}

// ------------------------------------------------------------------------------ stage B driver
__device__ __forceinline__ void stream_block(lz4amd_gsrc src, uint32_t csize, lz4amd_gdst dst, uint32_t prefix,
                                             const SeqRec* rectab, uint32_t nseq, uint32_t total_real,
                                             char* smem, uint64_t* prof) {
    const uint32_t tid = threadIdx.x;
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    uint8_t* cr = (uint8_t*)(smem + kOffCr);
    SeqRec* recs = (SeqRec*)(smem + kOffRecs);
    uint32_t* idx = (uint32_t*)(smem + kOffIdx);
    const uint32_t total = total_real + kBias;            // positions are biased by kBias from here on
    const uint32_t nreg = (uint32_t)(((uint64_t)total + kRegion - 1) >> kRegionShift);

    // uniform state (every thread computes the same values)
    uint32_t rec_head = 0, rec_tail = 0, out_emit = kBias, r_next = kBias >> kRegionShift;
    uint32_t cr_lo = 0, cr_hi = 0;
    uint32_t myfin = (kBias >> kRegionShift) / kDecWaves;                  // the history regions count as done
    uint32_t R = wave_id() + myfin * kDecWaves;           // per-wave copy cursor
    uint64_t t_emit = 0, t_copy = 0, t0 = 0;

    if (tid < kSlots) { DoneEnt e; e.mask = 0; e.tag = tid; e.pad = 0; ((DoneEnt*)(smem + kOffBits))[tid] = e; }
    if (tid < kDecWaves) ((uint32_t*)(smem + kOffFin))[tid] = myfin;
    // history before dst (linked blocks, lz4.c:2719 usingDict prefix mode) -> ring
    {
        uint8_t* ring = (uint8_t*)(smem + kOffRing);
        const uint32_t lo = kBias - prefix;
        for (uint32_t v = (lo & ~15u) + 16 * tid; v < kBias; v += 16 * kDecThreads) {
            U32x4 g; g[0] = g[1] = g[2] = g[3] = 0;
#pragma nounroll
            for (uint32_t i = 0; i < 16; i++) if (v + i >= lo) chunk_set_byte(g, i, (uint32_t)(dst - (kBias - (v + i)))[0]);
            const uint32_t a = ring_addr(v);
            *(U32x4*)(ring + a) = g;
            if (a < kRingPad) *(U32x4*)(ring + kRingBytes + a) = g;
        }
    }
    // what the prefetch registers hold: window granules [pf_lo, pf_hi) and table records rec_head+tid
    uint32_t cr_lo_n = 0, cr_hi_n = csize < kCrBytes ? csize : kCrBytes;
    uint32_t pf_lo = 0, pf_hi = cr_hi_n;
    uint32_t ncand = nseq + 1 < kRecCap - 1 ? nseq + 1 : kRecCap - 1;      // table rows [rec_head, rec_head+ncand), incl. a sentinel
    U32x4 pf0, pf1; pf0[0] = pf0[1] = pf0[2] = pf0[3] = 0; pf1 = pf0;
    SeqRec prec; prec.outpos = prec.litpos = prec.ll = prec.off = 0;
    {
        const uint32_t P0 = pf_lo + 16 * tid, P1 = P0 + 16 * kDecThreads;
        if (P0 < pf_hi) pf0 = load_granule(src, csize, P0);
        if (P1 < pf_hi) pf1 = load_granule(src, csize, P1);
        if (tid < ncand) prec = rectab[tid];
    }
    while (r_next < nreg) {
        if (prof && tid == 0) t0 = clock_ticks();
        // ---- commit the prefetched window granules; the slots of [old cr_lo, new cr_lo) are free
        {
            const uint32_t P0 = pf_lo + 16 * tid, P1 = P0 + 16 * kDecThreads;
            if (P0 < pf_hi) *(U32x4*)(cr + (P0 & kCrMask)) = pf0;
            if (P1 < pf_hi) *(U32x4*)(cr + (P1 & kCrMask)) = pf1;
        }
        cr_lo = cr_lo_n; cr_hi = cr_hi_n;
        // ---- commit the prefetched records whose literals lie inside the window (litpos grows with the
        //      sequence number, so a prefix passes); the first one always goes (HBM literals if need be)
        const bool mine = tid < ncand && rec_head + tid < nseq;          // a real record, not the sentinel
        const int ok = mine && tid + 1 < ncand && (tid == 0 || prec.litpos + prec.ll <= cr_hi);   // row ncand-1 can only be the sentinel
        const uint32_t nacc = (uint32_t)__syncthreads_count(ok);
        if (tid <= nacc && tid < ncand) recs[(rec_head + tid) & kRecMask] = prec;      // row nacc = sentinel (real record later)
        if (tid == nacc && tid < ncand) misc[M_EMIT] = prec.outpos;
        __syncthreads();
        if (ncand) { rec_head += nacc; out_emit = misc[M_EMIT]; }
        // ---- INDEX: first record of every region the ring covers
        for (uint32_t j = rec_tail + tid; j < rec_head; j += kDecThreads) {
            const uint32_t o = recs[j & kRecMask].outpos, on = recs[(j + 1) & kRecMask].outpos;
            uint32_t g = (o + kRegion - 1) >> kRegionShift;
            if (g < r_next) g = r_next;
            for (; g <= r_next + kIdxCap && ((uint64_t)g << kRegionShift) < on; g++) idx[g & kIdxMask] = j;
        }
        uint32_t r_ready = rec_head >= nseq ? nreg : (out_emit >> kRegionShift);
        if (r_ready > r_next + kIdxCap) r_ready = r_next + kIdxCap;
        __syncthreads();
        // ---- plan the next iteration and issue its global loads (committed after the copy)
        const uint32_t x = r_ready << kRegionShift;
        const uint32_t rec_tail_n = (r_ready < nreg && x < out_emit) ? idx[r_ready & kIdxMask] : rec_head;
        uint32_t need;
        {
            const SeqRec r = recs[rec_tail_n & kRecMask];            // rec_tail_n == rec_head: the sentinel row = next record
            uint32_t d = (rec_tail_n < rec_head && x > r.outpos) ? x - r.outpos : 0; if (d > r.ll) d = r.ll;
            need = r.litpos + d; if (need > csize) need = csize;
        }
        cr_lo_n = need & ~15u; if (cr_lo_n < cr_lo) cr_lo_n = cr_lo;
        cr_hi_n = cr_lo_n + kCrBytes; if (cr_hi_n > csize) cr_hi_n = csize;
        pf_lo = cr_hi > cr_lo_n ? cr_hi : cr_lo_n; pf_hi = cr_hi_n;
        {
            const uint32_t room = kRecCap - 1 - (rec_head - rec_tail_n);  // ring rows free after this iteration
            const uint32_t left = nseq + 1 - rec_head;                     // table rows left, sentinel included
            ncand = room < left ? room : left; if (ncand > kDecThreads) ncand = kDecThreads;
            const uint32_t P0 = pf_lo + 16 * tid, P1 = P0 + 16 * kDecThreads;
            if (P0 < pf_hi) pf0 = load_granule(src, csize, P0);
            if (P1 < pf_hi) pf1 = load_granule(src, csize, P1);
            if (tid < ncand) prec = rectab[rec_head + tid];
        }
        if (prof && tid == 0) { const uint64_t t1 = clock_ticks(); t_emit += t1 - t0; t0 = t1; }
        // ---- COPY
        CopyCtx C; C.src = src; C.csize = csize; C.dst = dst; C.out_emit = out_emit; C.rec_head = rec_head;
        C.cr_lo = cr_lo; C.cr_hi = cr_hi; C.r_ready = r_ready;
        copy_regions(C, smem, R, myfin);
        __syncthreads();
        if (prof && tid == 0) t_copy += clock_ticks() - t0;
        r_next = r_ready; rec_tail = rec_tail_n;
    }
    if (prof && tid == 0) { prof[2] = t_emit; prof[3] = t_copy; }
}

__device__ __forceinline__ void decode_one_block(const DecBatch& P, uint32_t b, char* smem) {
    const uint32_t tid = threadIdx.x;
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);

    const lz4amd_gsrc src = LZ4AMD_TO_GSRC(P.src[b]);
    const lz4amd_gdst dst = LZ4AMD_TO_GDST(P.dst[b]);
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

    SeqRec* rectab = (SeqRec*)(P.scratch + (uint64_t)blockIdx.x * P.scratch_stride);
    uint64_t* prof = P.prof ? P.prof + (uint64_t)blockIdx.x * 8 : nullptr;
    if (prof && tid == 0) prof[0] = clock_ticks();

    uint32_t nseq = 0, total = 0;
    uint32_t prefix = P.prefix ? (uint32_t)P.prefix[b] : 0u; if (prefix > kBias) prefix = kBias;
    if (!preparse_block(src, csize, cap, prefix, rectab, smem, nseq, total, prof)) {
        if (tid == 0) P.result[b] = err_at(misc[M_ERR]);
        return;
    }
    if (prof && tid == 0) prof[1] = clock_ticks();
    __syncthreads();            // record table visible to the whole workgroup; stage A's LDS is dead
    stream_block(src, csize, dst, prefix, rectab, nseq, total, smem, prof);
    if (tid == 0) P.result[b] = (int32_t)total;
    if (prof && tid == 0) prof[4] = clock_ticks();
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

} } // namespace lz4amd::v1
