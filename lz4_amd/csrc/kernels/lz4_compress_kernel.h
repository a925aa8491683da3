// lz4_compress_kernel.h -- batched LZ4 block compression (fast / "default" level) for gfx950.
//
// Replaces, for a whole batch of independent blocks resident in HBM, what the reference does per
// block in LZ4_compress_default (lib/lz4.c:1472 -> LZ4_compress_fast_extState lz4.c:1382 ->
// LZ4_compress_generic_validated lz4.c:930-1338): greedy single-candidate hash-table LZ77 parse,
// emitted as one legal LZ4 block (doc/lz4_Block_format.md; end-of-block rules MFLIMIT /
// LASTLITERALS lz4.c:242-263, 963-964).  The bytes differ from the CPU library's (the parse is
// done 64 positions at a time) but decode identically with any LZ4 decoder; the table-size policy
// mirrors the reference (13-bit 4-byte hash for blocks < 64 KB+11, lz4.c:1389; 12 bits above) so
// the ratio stays within a few tenths of a percent of it on the datagen inputs.
//
// Not a port: the reference is one serial loop over one block.  Here a block is cut in sub-chunks
// of `sub_bytes` (64 KB) and the work is three launches over ALL sub-chunks of ALL blocks:
//
//   K_match   one wave per sub-chunk.  The wave seeds a private LDS hash table with the 64 KB that
//             precede its sub-chunk (so the window is not lost at the cut), then slides a
//             64-position window: every lane hashes its position, probes and updates the table,
//             verifies its candidate (4 bytes); found matches are taken greedily in position order
//             (ballot + ctz), extended backwards over pending literals and forwards by a
//             wave-wide 256-byte compare.  Output: 8-byte (literals, match, offset) records plus
//             the encoded size of the sub-chunk.  No bytes of the final stream are written yet.
//   K_offsets one wave per block: prefix-sum of sub-chunk sizes, literal-run carry across
//             sub-chunk cuts, capacity check (0 = does not fit, lz4.c:1114-1117 semantics),
//             final literal run header.
//   K_emit    one wave per sub-chunk: wave prefix-sums over 64 sequences at a time give every
//             token its byte offset; lanes write their token/length/offset fields, then the wave
//             copies the literal runs (coalesced byte lanes).  Literals that are carried into a
//             later sequence are still copied by the wave that owns their source bytes, so
//             incompressible blocks are copied by all waves in parallel.
//
// HBM traffic: source read once by K_match (+ the 64 KB seeding overlap, L2-resident), literals
// re-read by K_emit (mostly L2/MALL hits), compressed stream written once.  No MFMA.
#pragma once
#include "lz4_common.h"
#include "../lz4amd_params.h"

namespace lz4amd {

struct alignas(8) MatchRec { uint32_t ll; uint32_t mo; };   // mo = offset | (matchlen-4) << 16

enum : uint32_t {
    kSubBytes = 64u << 10,             // default sub-chunk
    kMaxRecsPerSub = (kSubBytes / 4) + 8,
    kSmallBlockLimit = 65536 + 11,     // lz4.c:710 LZ4_64Klimit
    kMatchLdsBytes = 16384,            // 8192 x u16 (13-bit) or 4096 x u16 (12-bit)
    kNoOutput = 0xFFFFFFFFu,
};

using CompBatch = ::lz4amd_comp_params;   // argument block (lz4amd_params.h)

__device__ __forceinline__ uint32_t len_ext_bytes(uint32_t len_minus_nibble_base) {
    // bytes needed after the token for a length field whose value is >= 15 (block format doc)
    return 1 + len_minus_nibble_base / 255;
}
__device__ __forceinline__ uint32_t enc_size(uint32_t ll, uint32_t mlm4) {
    uint32_t s = 1 + ll + 2;
    if (ll >= 15) s += len_ext_bytes(ll - 15);
    if (mlm4 >= 15) s += len_ext_bytes(mlm4 - 15);
    return s;
}
__device__ __forceinline__ uint8_t* put_len_ext(uint8_t* p, uint32_t rest) {
    while (rest >= 255) { *p++ = 255; rest -= 255; }
    *p++ = (uint8_t)rest;
    return p;
}

// ------------------------------------------------------------------------------ K_match
// hash flavours of the reference (lz4.c:777-795): 4-byte multiplicative hash into 13 bits for
// inputs below 64 KB + 11, 5-byte hash into 12 bits above.  The 5-byte hash matters for more than
// parity: candidates that agree in only 4 bytes save one byte at best and triple the number of
// sequences the decoder has to walk.
__device__ __forceinline__ uint32_t hash_small(uint32_t v) { return (v * 2654435761u) >> (32 - 13); }
__device__ __forceinline__ uint32_t hash_large(uint64_t v8) { return (uint32_t)(((v8 << 24) * 889523592379ull) >> (64 - 12)); }

// The reference probes a literal run with a growing stride: 64 probes at stride 1, 64 at stride 2,
// ... restarting after every match (lz4.c:1022-1053, LZ4_skipTrigger = 6).  Positions it jumps
// over are neither tested nor INDEXED, and with a 4096-entry table that matters: indexing every
// byte of a long literal run evicts the older entries the next match needs (ratio -2.5 % on the
// datagen inputs).  x = distance from the start of the run; true if the reference would probe it.
__device__ __forceinline__ bool is_probe_position(uint32_t x) {
    if (x < 64) return true;
    uint32_t j = (uint32_t)((sqrtf(1.0f + (float)x * 0.125f) - 1.0f) * 0.5f);   // 32 j (j+1) <= x
    while (32u * j * (j + 1) > x) j--;
    while (32u * (j + 1) * (j + 2) <= x) j++;
    const uint32_t k = x - 32u * j * (j + 1), d = j + 1;
    const uint32_t t = (uint32_t)((float)k / (float)d + 0.5f);
    return t * d == k;
}

// number of equal leading bytes (0..8) of two 8-byte little-endian words
__device__ __forceinline__ uint32_t equal_bytes8(uint64_t x, uint64_t y) {
    const uint64_t d = x ^ y;
    if (d == 0) return 8;
    const uint32_t lo = (uint32_t)d;
    return lo ? (uint32_t)(__ffs((int)lo) - 1) >> 3 : 4 + ((uint32_t)(__ffs((int)(uint32_t)(d >> 32)) - 1) >> 3);
}

__device__ __forceinline__ void match_subchunk_body(const CompBatch& P) {
    LZ4AMD_DYN_LDS(smem);
    uint16_t* tab = (uint16_t*)smem;
    const uint32_t lane = lane_id();
    const uint32_t k = blockIdx.x;                    // one wave per sub-chunk
    const uint32_t b = P.sub_block[k];
    const uint8_t* __restrict__ src = P.src[b];
    const int32_t n_i = P.src_size[b];
    const uint32_t n = n_i > 0 ? (uint32_t)n_i : 0;
    const uint32_t cs = (k - P.blk_sub0[b]) * P.sub_bytes;
    uint32_t ce = cs + P.sub_bytes; if (ce > n || ce < cs) ce = n;
    MatchRec* recs = (MatchRec*)P.recs + (uint64_t)k * P.recs_per_sub;

    const bool small = n < kSmallBlockLimit;
    // -- clear table (16-byte stores)
    {
        U32x4 z; z[0] = z[1] = z[2] = z[3] = 0;
        const uint32_t n16 = small ? 1024u : 512u;
        for (uint32_t i = lane; i < n16; i += 64) ((U32x4*)tab)[i] = z;
    }
    wave_lds_fence();
    uint32_t nseq = 0, enc = 0, anchor = cs;
    // positions that may start a match: q <= n - 12; matches end <= n - 5 and <= ce
    if (n >= kMfLimit + 1 && cs <= n - kMfLimit) {
        const uint32_t last_q = n - kMfLimit;                  // inclusive; q + 8 <= n - 4 holds for all probes
        uint32_t mlimit = n - kLastLiterals; if (mlimit > ce) mlimit = ce;
        // -- seed with the window that precedes the sub-chunk
        const uint32_t low = cs > kMaxDistance ? cs - kMaxDistance : 0;
        for (uint32_t p = low; p < cs; p += 64) {
            const uint32_t q = p + lane;
            if (q < cs) {
                const uint64_t v8 = ld_u64(src + q);
                tab[small ? hash_small((uint32_t)v8) : hash_large(v8)] = (uint16_t)q;
            }
        }
        wave_lds_fence();
        uint32_t p = cs, cur = cs;                             // cur: first position not yet covered
        uint32_t run0 = cs;                                    // start of the current literal run's probe schedule
        uint64_t vpre = 0; uint32_t ppre = kNoOutput;          // source words fetched one window ahead
        while (p < ce && p <= last_q) {
            const uint32_t q = p + lane;
            const bool valid = q < ce && q <= last_q;
            uint64_t v8 = 0;
            if (ppre == p) v8 = vpre; else if (valid) v8 = ld_u64(src + q);
            {   // prefetch the next window (discarded when a long match jumps over it)
                const uint32_t qn = q + 64;
                ppre = p + 64;
                vpre = (qn < ce && qn <= last_q) ? ld_u64(src + qn) : 0ull;
            }
            const uint32_t v = (uint32_t)v8;
            uint32_t h = 0, c1 = 0, c2 = 0;
            bool ok1 = false;
            if (valid) {
                h = small ? hash_small(v) : hash_large(v8);
                const uint32_t d1 = (q - tab[h]) & 0xFFFFu;
                c1 = q - d1;
                ok1 = d1 != 0 && d1 <= q && c1 >= low;
            }
            wave_lds_fence();                              // every lane probes before any lane inserts
            // candidates closer than one window are invisible to the table probe (the window's own
            // positions are inserted after the probe): catch the short periods 1..4 (runs, 16/32-bit
            // patterns) by comparing with the neighbouring lanes' bytes instead
            const bool f1 = ok1 && (uint32_t)ld_u32(src + c1) == v;
            bool f2 = false;
#pragma unroll
            for (uint32_t d = 4; d >= 1; d--) {
                const uint32_t vd = __shfl_up(v, d);
                const bool vld = (bool)__shfl_up((int)valid, d);
                if (lane >= d && valid && vld && vd == v) { c2 = q - d; f2 = true; }
            }
            f2 = f2 && !f1;
            const uint32_t cand = f1 ? c1 : c2;
            const bool e_old = valid && q >= run0 && is_probe_position(q - run0);
            const unsigned long long em = __ballot(e_old);
            uint32_t firstcur = kNoOutput;              // end of the first match taken in this window
            unsigned long long m = __ballot(f1 || f2);
            // window positions swallowed by matches are not indexed, except cur-2 (lz4.c:1236-1242);
            // a window that starts inside the previous match (p = cur-2) only indexes its lane 0
            unsigned long long covered = 0;
            if (cur > p) covered = ((cur - p >= 64) ? ~0ull : ((1ull << (cur - p)) - 1)) & ~(1ull << (cur - 2 - p));
            while (m) {
                const uint32_t l = (uint32_t)__ffsll((long long)m) - 1;
                m &= m - 1;
                uint32_t qm = p + l;
                const uint32_t qfound = qm;
                if (qm < cur) continue;
                if (firstcur == kNoOutput && !((em >> l) & 1)) continue;     // not on the probe schedule
                uint32_t cm = (uint32_t)__shfl((int)cand, (int)l);
                if (qm + kMinMatch > mlimit) continue;          // the 4 verified bytes cross the cut
                // -- forward (lz4.c:680-703 LZ4_count) and backward (lz4.c:1105-1109) extension
                //    in one memory round trip: 8 bytes per lane forward, 1 byte per lane backward
                uint32_t room = qm - anchor; if (cm < room) room = cm;
                uint32_t ml = kMinMatch, back = 0;
                bool fwd_done = false, back_done = room == 0;
                uint32_t fbase = kMinMatch, bbase = 0;
                while (!fwd_done || !back_done) {
                    uint32_t same = 8; bool beq = true;
                    if (!fwd_done) {
                        const uint32_t a = qm + fbase + 8 * lane, c = cm + fbase + 8 * lane;
                        if (a >= mlimit) same = 0;
                        else if (a + 8 <= n) { same = equal_bytes8(ld_u64(src + a), ld_u64(src + c)); if (same > mlimit - a) same = mlimit - a; }
                        else { same = 0; while (a + same < mlimit && src[a + same] == src[c + same]) same++; }
                    }
                    if (!back_done) {
                        const uint32_t i = bbase + lane;
                        beq = i < room && src[qm - 1 - i] == src[cm - 1 - i];
                    }
                    if (!fwd_done) {
                        const unsigned long long brk = __ballot(same < 8);
                        if (brk) {
                            const uint32_t fl = (uint32_t)__ffsll((long long)brk) - 1;
                            ml = fbase + 8 * fl + (uint32_t)__shfl((int)same, (int)fl);
                            fwd_done = true;
                        } else fbase += 512;
                    }
                    if (!back_done) {
                        const unsigned long long ne = __ballot(!beq);
                        if (ne) { back = bbase + (uint32_t)__ffsll((long long)ne) - 1; back_done = true; }
                        else bbase += 64;
                    }
                }
                qm -= back; cm -= back; ml += back;
                const uint32_t ll = qm - anchor;
                if (lane == 0) { MatchRec r; r.ll = ll; r.mo = (qm - cm) | ((ml - kMinMatch) << 16); recs[nseq] = r; }
                enc += enc_size(ll, ml - kMinMatch);
                nseq++;
                anchor = cur = qm + ml;
                run0 = cur;
                if (firstcur == kNoOutput) firstcur = cur;
                {   // window lanes inside [qm, cur) are not indexed, except cur-2 (lz4.c:1236-1242)
                    const uint32_t a0 = qfound - p;     // positions passed over before the hit stay indexed
                    const uint32_t a1 = cur - p < 64 ? cur - p : 64;
                    if (a1 > a0) covered |= ((a1 - a0 >= 64) ? ~0ull : ((1ull << (a1 - a0)) - 1)) << a0;
                    if (cur - 2 >= p && cur - 2 < p + 64) covered &= ~(1ull << (cur - 2 - p));
                }
            }
            if (valid && !((covered >> lane) & 1) && (q >= firstcur || q < run0 || e_old)) tab[h] = (uint16_t)q;
            p = (cur > p + 66) ? cur - 2 : p + 64;
        }
    }
    if (lane == 0) {
        P.sub_n[k] = nseq;
        P.sub_enc[k] = enc;
        P.sub_tail[k] = ce - anchor;
    }
}

// ---------------------------------------------------------------------------- K_offsets
// one wave per block; lanes stride over the block's sub-chunks in order (serial carry chain,
// <= a few hundred sub-chunks for the block sizes of interest; 32768 for a 2 GB block).
__device__ __forceinline__ void offsets_body(const CompBatch& P) {
    const uint32_t b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const int32_t n_i = P.src_size[b];
    const int32_t cap_i = P.dst_cap[b];
    const uint32_t s0 = P.blk_sub0[b], s1 = P.blk_sub0[b + 1];
    if (n_i < 0 || (uint32_t)n_i > 0x7E000000u || cap_i <= 0 || P.dst[b] == nullptr) { P.result[b] = 0; return; }
    uint8_t* dst = P.dst[b];
    if (n_i == 0) { dst[0] = 0; P.result[b] = 1; return; }         // lz4.c:1361-1371
    const uint32_t n = (uint32_t)n_i, cap = (uint32_t)cap_i;
    // pass 1: sizes
    uint64_t out = 0; uint32_t carry = 0;
    for (uint32_t k = s0; k < s1; k++) {
        const uint32_t nk = P.sub_n[k];
        if (nk) {
            // first sequence of this sub-chunk absorbs the carried literals
            const uint32_t ll0 = ((const MatchRec*)P.recs)[(uint64_t)k * P.recs_per_sub].ll;
            uint32_t delta = carry;
            const uint32_t a = ll0 >= 15 ? len_ext_bytes(ll0 - 15) : 0;
            const uint32_t bb = (ll0 + carry) >= 15 ? len_ext_bytes(ll0 + carry - 15) : 0;
            delta += bb - a;
            P.sub_out[k] = (uint32_t)out;
            P.sub_carry[k] = carry;
            out += (uint64_t)P.sub_enc[k] + delta;
            carry = P.sub_tail[k];
        } else {
            P.sub_out[k] = (uint32_t)out; P.sub_carry[k] = 0;
            carry += P.sub_tail[k];
        }
    }
    const uint32_t last_run = carry;
    const uint64_t total = out + 1 + (last_run >= 15 ? len_ext_bytes(last_run - 15) : 0) + last_run;
    if (total > cap) {                                   // does not fit: 0 (lz4.c:1116,1210,1314)
        P.result[b] = 0;
        for (uint32_t k = s0; k < s1; k++) P.sub_out[k] = kNoOutput;
        return;
    }
    // pass 2: destinations of tail literals.  A tail run belongs to the next sequence that
    // exists (its literals end right before that sequence's offset field) or to the last run.
    {
        uint8_t* p = dst + out;
        if (last_run >= 15) { *p++ = 0xF0; p = put_len_ext(p, last_run - 15); }
        else *p++ = (uint8_t)(last_run << 4);
        uint32_t lit_end = (uint32_t)(p - dst) + last_run;      // end of the pending literal area
        for (uint32_t k = s1; k-- > s0;) {
            const uint32_t tail = P.sub_tail[k];
            P.sub_tail_dst[k] = lit_end - tail;
            if (P.sub_n[k]) {
                // earlier tails feed this sub-chunk's first sequence: its literal area ends
                // where its own (non-carried) ll0 literals begin
                const uint32_t ll0 = ((const MatchRec*)P.recs)[(uint64_t)k * P.recs_per_sub].ll;
                const uint32_t c = P.sub_carry[k];
                const uint32_t hdr = 1 + ((ll0 + c) >= 15 ? len_ext_bytes(ll0 + c - 15) : 0);
                lit_end = P.sub_out[k] + hdr + c;
            } else {
                lit_end -= tail;
            }
        }
    }
    (void)n;
    P.result[b] = (int32_t)total;
}

// ------------------------------------------------------------------------------- K_emit
__device__ __forceinline__ void wave_copy_bytes(uint8_t* __restrict__ d, const uint8_t* __restrict__ s, uint32_t n) {
    for (uint32_t i = lane_id(); i < n; i += 64) d[i] = s[i];
}

__device__ __forceinline__ void emit_subchunk_body(const CompBatch& P) {
    const uint32_t lane = lane_id();
    const uint32_t k = blockIdx.x;
    const uint32_t out0 = P.sub_out[k];
    if (out0 == kNoOutput) return;                      // block failed / nothing to do (uniform)
    const uint32_t b = P.sub_block[k];
    const uint8_t* __restrict__ src = P.src[b];
    uint8_t* __restrict__ dst = P.dst[b];
    const uint32_t n = (uint32_t)P.src_size[b];
    const uint32_t cs = (k - P.blk_sub0[b]) * P.sub_bytes;
    uint32_t ce = cs + P.sub_bytes; if (ce > n || ce < cs) ce = n;
    const uint32_t nk = P.sub_n[k];
    const uint32_t carry = P.sub_carry[k];
    const MatchRec* recs = (MatchRec*)P.recs + (uint64_t)k * P.recs_per_sub;

    uint32_t ipos = cs;            // source position of the next sequence's literals
    uint32_t opos = out0;          // dst position of the next sequence's token
    for (uint32_t base = 0; base < nk; base += 64) {
        const uint32_t i = base + lane;
        const bool have = i < nk;
        uint32_t ll = 0, mlm4 = 0, off = 0, extra = 0;
        if (have) { const MatchRec r = recs[i]; ll = r.ll; off = r.mo & 0xFFFFu; mlm4 = r.mo >> 16; }
        if (i == 0) extra = carry;                       // literals inherited from earlier sub-chunks
        const uint32_t e = have ? enc_size(ll + extra, mlm4) : 0;
        const uint32_t adv = have ? ll + mlm4 + kMinMatch : 0;
        const uint32_t e_incl = wave_incl_sum(e), a_incl = wave_incl_sum(adv);
        const uint32_t my_o = opos + e_incl - e;
        const uint32_t my_i = ipos + a_incl - adv;       // source pos of my own ll literals
        uint32_t lit_dst = 0;
        if (have) {
            uint8_t* p = dst + my_o;
            const uint32_t tl = ll + extra;
            const uint32_t tok_ll = tl >= 15 ? 15u : tl, tok_ml = mlm4 >= 15 ? 15u : mlm4;
            *p++ = (uint8_t)((tok_ll << 4) | tok_ml);
            if (tl >= 15) p = put_len_ext(p, tl - 15);
            lit_dst = (uint32_t)(p - dst) + extra;       // my own literals follow the carried ones
            p += tl;
            p[0] = (uint8_t)off; p[1] = (uint8_t)(off >> 8); p += 2;
            if (mlm4 >= 15) p = put_len_ext(p, mlm4 - 15);
        }
        // literal runs, one sequence at a time, all lanes copying
        const uint32_t cnt = nk - base < 64 ? nk - base : 64;
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t jl = (uint32_t)__shfl((int)ll, (int)j);
            if (jl == 0) continue;
            const uint32_t jd = (uint32_t)__shfl((int)lit_dst, (int)j);
            const uint32_t js = (uint32_t)__shfl((int)my_i, (int)j);
            wave_copy_bytes(dst + jd, src + js, jl);
        }
        opos += (uint32_t)__shfl((int)e_incl, 63);
        ipos += (uint32_t)__shfl((int)a_incl, 63);
    }
    // tail literals of this sub-chunk (carried into a later sequence or the final run)
    const uint32_t tail = P.sub_tail[k];
    if (tail) wave_copy_bytes(dst + P.sub_tail_dst[k], src + (ce - tail), tail);
}

} // namespace lz4amd
