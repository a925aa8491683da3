// lz4_compress_kernel.h -- batched LZ4 block compression (fast / "default" level) for gfx950.
//
// Replaces, for a whole batch of independent blocks resident in HBM, what the reference does per
// block in LZ4_compress_default (lib/lz4.c:1472 -> LZ4_compress_fast_extState lz4.c:1382 ->
// LZ4_compress_generic_validated lz4.c:930-1338): greedy single-candidate hash-table LZ77 parse,
// emitted as one legal LZ4 block (doc/lz4_Block_format.md; end-of-block rules MFLIMIT /
// LASTLITERALS lz4.c:242-263, 963-964).  The bytes differ from the CPU library's but decode
// identically with any LZ4 decoder, and the compressed size stays within about 1 % of the
// reference's on the datagen inputs (tests/test_gpu_parity.py asserts the +-3 % window).
//
// Not a port: the reference is one serial loop over one block, whose hash table changes after
// every probe.  Here ONE 1024-thread workgroup (16 waves, one CU, ~150 KB of its LDS) streams
// through a block in TILES of 8 KB, and everything the inner loops touch lives in LDS:
//
//   source ring   100 KB of the block around the tile (the 64 KB LZ4 window + the tile + the
//                 prefetched next tile), filled with coalesced 16-byte loads; candidates are
//                 verified and matches extended against it, never against HBM;
//   hash table    8192 x u32 (5-byte multiplicative hash of the reference, lz4.c:785-795, 13 bits),
//                 FROZEN while a tile is parsed: every position of the tile probes the state left
//                 by the previous tiles, and the tile's own positions are inserted afterwards with
//                 atomicMax (order independent, so the output is deterministic).  That decouples
//                 match FINDING from the parse order, which is what makes the tile parallel:
//   strips        the tile is cut in 16 strips of 512 bytes, one wave each.  A wave slides a
//                 64-position window over its strip: every lane hashes its position, probes,
//                 measures up to 24 matching bytes forwards and 8 backwards (over
//                 literals that may still be pending) on its own; found matches
//                 are taken greedily in position order (ballot + ctz), long ones are extended by a
//                 wave-wide 512-byte compare.  Matches end at the strip's end; literals pending at
//                 a strip's end are carried into the next sequence (of any later strip or tile).
//   offsets       after a barrier one thread turns the strips' encoded sizes into output offsets
//                 (running across tiles), while the other waves insert the tile into the table;
//   emit          each wave writes the sequences of its strip: lanes place their token / length /
//                 offset bytes by a wave prefix sum, then the wave copies the literal runs out of
//                 the source ring (HBM for runs that started more than a window ago).
//
// HBM traffic per block: source read once (plus the final literal run if it is longer than the
// ring), compressed stream written once.  No scratch, no second kernel.  No MFMA: byte shuffling.
#pragma once
#include "lz4_common.h"
#include "../lz4amd_params.h"

namespace lz4amd {

using CompBatch = ::lz4amd_comp_params;   // argument block (lz4amd_params.h)

struct alignas(8) MatchRec { uint32_t ll; uint32_t mo; };   // mo = offset | (matchlen-4) << 16

enum : uint32_t {
    kCmpThreads = 1024,
    kCmpWaves = kCmpThreads / 64,
    kTileMax = 8192,                   // blocks >= 64 KB + 11 (which are probed at every second position)
    kTileMaxSmall = 2048,              // smaller blocks (4-byte hash, like the reference: lz4.c:1389)
    kTileMin = 1024,
    kStripMin = 256,
    kSrcRing = 100u << 10,
    kSrcPad = 32,                      // mirror of the ring's first bytes: unaligned reads never wrap
    kHashBits = 13,
    kRecsPerStrip = 64,                // matches a strip may take (the rest of it becomes literals)
    kLaneLenCap = 24,                  // match bytes a lane measures on its own
    kShortRun = 16,                    // literal runs up to this long are copied by the sequence's own lane (4 / 8 / 32 measured: no better)
    kMaxInput = 0x7E000000u,           // lz4.h:214 LZ4_MAX_INPUT_SIZE
    kSmallBlockLimit = 65536 + 11,     // lz4.c:710 LZ4_64Klimit
    kStripFields = 9,
};
// LDS carve-up (bytes)
enum : uint32_t {
    kCOffMisc = 0,                                        // u32[32]
    kCOffStrip = kCOffMisc + 32 * 4,                      // u32[2][kStripFields][16] per-strip summaries (two tiles in flight)
    kCOffTab = kCOffStrip + 2 * kStripFields * kCmpWaves * 4,   // u32[1 << kHashBits]
    kCOffRecs = kCOffTab + (4u << kHashBits),             // MatchRec[2][kCmpWaves][kRecsPerStrip]
    kCOffEnds = kCOffRecs + 2 * kCmpWaves * kRecsPerStrip * 8,      // u16[2][kCmpWaves][kRecsPerStrip] where a record's match ends (from the strip's start)
    kCOffEncp = kCOffEnds + 2 * kCmpWaves * kRecsPerStrip * 2,      // u16[2][kCmpWaves][kRecsPerStrip] encoded bytes of the strip's records before it
    kCOffRing = kCOffEncp + 2 * kCmpWaves * kRecsPerStrip * 2,
    kCmpLdsBytes = kCOffRing + kSrcRing + kSrcPad,
};
enum : uint32_t { CM_BLOCK = 0, CM_OUT = 1, CM_CARRY = 2, CM_FAIL = 3, CM_READY = 4 };      // CM_READY: tiles whose output offsets are fixed
enum : uint32_t { S_N = 0, S_ENC = 1, S_LL0 = 2, S_TAIL = 3, S_OUT = 4, S_CARRY = 5,
                  S_END = 6,      // where the strip's last match ends when it runs past the strip (else 0)
                  S_FIRST = 7,    // first record that is emitted (the ones before it were covered by an earlier strip's match)
                  S_P = 8 };      // first source position the strip emits

__device__ __forceinline__ uint32_t len_ext_bytes(uint32_t len_minus_nibble_base) {
    // bytes needed after the token for a length field whose value is >= 15 (block format doc)
    return 1 + len_minus_nibble_base / 255;
}
__device__ __forceinline__ uint32_t lit_hdr_ext(uint32_t ll) { return ll >= 15 ? len_ext_bytes(ll - 15) : 0; }
__device__ __forceinline__ uint32_t enc_size(uint32_t ll, uint32_t mlm4) {
    uint32_t s = 1 + ll + 2;
    if (ll >= 15) s += len_ext_bytes(ll - 15);
    if (mlm4 >= 15) s += len_ext_bytes(mlm4 - 15);
    return s;
}
template <class Ptr> __device__ __forceinline__ Ptr put_len_ext(Ptr p, uint32_t rest) {
    while (rest >= 255) { *p++ = 255; rest -= 255; }
    *p++ = (uint8_t)rest;
    return p;
}

// 13-bit hashes of the 4 (blocks < 64 KB + 11, lz4.c:777-783) or 5 (lz4.c:785-795) bytes at a position.
// The 5-byte one is not the reference's 64-bit multiply (four quarter-rate 32-bit multiplies on this
// chip) but two 32-bit multiplicative hashes of the same bytes added up; any well mixed function of
// the 5 bytes gives the same matches up to table collisions.
__device__ __forceinline__ uint32_t hash_pos32(uint32_t lo, uint32_t hi, bool small) {
    const uint32_t h4 = lo * 2654435761u;
    return (small ? h4 : h4 + (hi & 0xFFu) * 0x85EBCA77u) >> (32 - kHashBits);
}
__device__ __forceinline__ uint32_t hash_pos(uint64_t v8, bool small) { return hash_pos32((uint32_t)v8, (uint32_t)(v8 >> 32), small); }

// number of equal leading bytes (0..8) of two 8-byte little-endian words
__device__ __forceinline__ uint32_t equal_bytes8(uint64_t x, uint64_t y) {
    const uint64_t d = x ^ y;
    if (d == 0) return 8;
    const uint32_t lo = (uint32_t)d;
    return lo ? (uint32_t)(__ffs((int)lo) - 1) >> 3 : 4 + ((uint32_t)(__ffs((int)(uint32_t)(d >> 32)) - 1) >> 3);
}

// ring offset of block position pos
__device__ __forceinline__ uint32_t src_ring_off(uint32_t pos) { return pos % kSrcRing; }
// offset `d` bytes before / after ring offset o (d < kSrcRing)
__device__ __forceinline__ uint32_t ring_back(uint32_t o, uint32_t d) { return o >= d ? o - d : o + kSrcRing - d; }
__device__ __forceinline__ uint32_t ring_fwd(uint32_t o, uint32_t d) { const uint32_t x = o + d; return x >= kSrcRing ? x - kSrcRing : x; }
// 8 bytes at ring offset o, any alignment: two ALIGNED 8-byte reads and a funnel shift (a misaligned
// ds_read_b64 costs about five aligned ones on gfx950; the pad covers the read past the ring's end)
__device__ __forceinline__ uint64_t funnel8(uint64_t lo, uint64_t hi, uint32_t byte_shift) {
    const uint32_t s = byte_shift * 8;
    return s ? (lo >> s) | (hi << (64 - s)) : lo;
}
__device__ __forceinline__ uint64_t ring_ld8(const uint8_t* ring, uint32_t o) {
    const uint64_t* a = (const uint64_t*)(ring + (o & ~7u));
    return funnel8(a[0], a[1], o & 7);
}

// number of equal leading bytes (0..8) of two 8-byte strings given as dword pairs
__device__ __forceinline__ uint32_t equal_bytes8_32(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1) {
    const uint32_t x = a0 ^ b0, y = a1 ^ b1;
    return x ? (uint32_t)(__ffs((int)x) - 1) >> 3 : (y ? 4 + ((uint32_t)(__ffs((int)y) - 1) >> 3) : 8u);
}
// The 32 bytes [pos-8, pos+24) around ring offset o as eight dwords (b0 b1 | f0..f5): nine aligned
// dword reads at constant offsets from one address and eight v_alignbyte.  (A window that starts before
// offset 0 wraps to the ring's end; the pad mirrors the ring's first 32 bytes, so no read wraps.)
struct Win32 { uint32_t b0, b1, f0, f1, f2, f3, f4, f5; };
__device__ __forceinline__ Win32 ring_window32(const uint8_t* ring, uint32_t o) {
    uint32_t a = ring_back(o, 8);
    const uint32_t sh = a & 3u;
    a &= ~3u;
    uint32_t d[9];
#pragma unroll
    for (uint32_t i = 0; i < 9; i++) d[i] = *(const uint32_t*)(ring + a + 4 * i);     // a <= kSrcRing - 4: the 36 bytes end inside the 32-byte pad
    Win32 w;
    w.b0 = align_bytes(d[1], d[0], sh); w.b1 = align_bytes(d[2], d[1], sh);
    w.f0 = align_bytes(d[3], d[2], sh); w.f1 = align_bytes(d[4], d[3], sh); w.f2 = align_bytes(d[5], d[4], sh);
    w.f3 = align_bytes(d[6], d[5], sh); w.f4 = align_bytes(d[7], d[6], sh); w.f5 = align_bytes(d[8], d[7], sh);
    return w;
}

// 16 source bytes at block position P (multiple of 16) -> ring
__device__ __forceinline__ void ring_commit16(uint8_t* ring, uint32_t P, const U32x4& v) {
    const uint32_t o = src_ring_off(P);
    *(U32x4*)(ring + o) = v;
    if (o < kSrcPad) *(U32x4*)(ring + kSrcRing + o) = v;
}
__device__ __forceinline__ U32x4 load_src16(lz4amd_gsrc src, uint32_t n, uint32_t P) {
    if (P + 16 <= n) return ld_global16(src + P);
    uint32_t a = 0, b = 0, c = 0, d = 0;
#pragma nounroll
    for (uint32_t i = 0; i < 16 && P + i < n; i++) {
        const uint32_t v = (uint32_t)src[P + i] << ((i & 3) * 8), k = i >> 2;
        a |= k == 0 ? v : 0; b |= k == 1 ? v : 0; c |= k == 2 ? v : 0; d |= k == 3 ? v : 0;
    }
    U32x4 r; r[0] = a; r[1] = b; r[2] = c; r[3] = d;
    return r;
}

// tile geometry at block position t0: small tiles first, so that small blocks (and the start of
// every block) are not parsed against an empty table for long
__device__ __forceinline__ void tile_geometry(uint32_t t0, bool small, uint32_t& tile_len, uint32_t& strip_len) {
    const uint32_t tmax = small ? kTileMaxSmall : kTileMax;
    uint32_t t = kTileMin;
    while (t < tmax && t * 4 <= t0) t <<= 1;
    tile_len = t;
    strip_len = t / kCmpWaves; if (strip_len < kStripMin) strip_len = kStripMin;
}

// ------------------------------------------------------------------------------ match (one strip)
__device__ __forceinline__ void match_strip(const uint8_t* ring, const uint32_t* tab, MatchRec* recs, uint16_t* ends, uint16_t* encp, uint32_t* strip,
                                            uint32_t w, uint32_t n, uint32_t cs, uint32_t ce, uint32_t tend) {
    const uint32_t lane = lane_id();
    const bool small = n < kSmallBlockLimit;
    uint32_t nseq = 0, enc = 0, ll0 = 0, anchor = cs;
    // positions that may start a match: q <= n - 12; matches end <= n - 5 and <= ce
    if (n >= kMfLimit + 1 && cs <= n - kMfLimit) {
        const uint32_t last_q = n - kMfLimit;                  // inclusive; q + 8 <= n - 4 holds for all probes
        // a match may run past the strip up to the tile's end: the strips it covers give way (resolve_overruns)
        uint32_t mlimit = n - kLastLiterals; if (mlimit > tend) mlimit = tend;
        const uint32_t cs_off = src_ring_off(cs);
        // big blocks probe every second position (the backward extension recovers the odd starts):
        // half the work for about 4 % of the matches, which the larger table more than pays for (every 4th position
        // was simulated: +5 % size at P60, +15 % at P90)
        const uint32_t sh = small ? 0u : 1u, span = 64u << sh, smask = (1u << sh) - 1u;
        uint32_t p = cs, cur = cs;                             // cur: first position not yet covered
        while (p < ce && p <= last_q && nseq < kRecsPerStrip) {
            const uint32_t q = p + (lane << sh);
            const bool valid = q < ce && q <= last_q && q != 0;
            const uint32_t qo = ring_fwd(cs_off, q - cs);
            uint32_t c = 0, len = 0, back = 0;
            bool f = false;
            if (valid) {
                // stage 1: my bytes q-8 .. q+24 (independent of the table)
                const Win32 v = ring_window32(ring, qo);
                // stage 2: the candidate
                c = tab[hash_pos32(v.f0, v.f1, small)];
                const uint32_t d = q - c;
                if (c < q && d <= kMaxDistance) {
                    // stage 3: its bytes c-8 .. c+24
                    const Win32 k = ring_window32(ring, ring_back(qo, d));
                    len = equal_bytes8_32(v.f0, v.f1, k.f0, k.f1);
                    if (len == 8) { len += equal_bytes8_32(v.f2, v.f3, k.f2, k.f3); if (len == 16) len += equal_bytes8_32(v.f4, v.f5, k.f4, k.f5); }
                    f = len >= kMinMatch;
                    // how far back, over literals that may still be pending (lz4.c:1105-1109)?
                    if (c >= 8) {
                        const uint32_t x1 = v.b1 ^ k.b1, x0 = v.b0 ^ k.b0;
                        back = x1 ? (uint32_t)__clz((int)x1) >> 3 : (x0 ? 4 + ((uint32_t)__clz((int)x0) >> 3) : 8u);
                    }
                }
            }
            // Which matches are taken is a serial question (a match starts where the previous one ended) but a
            // cheap one: the walk below only follows "first candidate at or after the end of the last one" and
            // tells every taken lane where its predecessor ended.  Everything per match - backward extension over
            // the pending literals, record, encoded size - is then done by the taken lanes side by side.
            uint32_t e = q + len;                                  // where my match ends (no lane-side length beyond the cap)
            if (e > mlimit) e = mlimit;
            uint32_t prev_end = 0;                                 // end of the match taken before mine (taken lanes only)
            unsigned long long m = __ballot(f && q + kMinMatch <= mlimit);
            unsigned long long taken = 0;
            uint32_t ntaken = 0, wend = anchor;                    // wend: end of the last match taken so far
            while (m) {
                // candidates that start inside what is already covered are out
                if (cur > p) { const uint32_t k = (cur - p + smask) >> sh; if (k >= 64) break; m &= ~0ull << k; if (!m) break; }
                const uint32_t l = (uint32_t)__ffsll((long long)m) - 1;
                m &= m - 1;
                uint32_t el = wave_readlane(e, l);
                const uint32_t qm = p + (l << sh);
                uint32_t ml = wave_readlane(len, l);
                if (ml >= kLaneLenCap && qm + ml < mlimit) {
                    // long match: wave-wide compare, 8 bytes per lane per trip (lz4.c:680-703 LZ4_count)
                    const uint32_t cm = wave_readlane(c, l);
                    const uint32_t qmo = ring_fwd(cs_off, qm - cs), cmo = ring_back(qmo, qm - cm);
                    for (;;) {
                        const uint32_t a = qm + ml + 8 * lane;
                        uint32_t same = 0;
                        if (a < mlimit) {
                            same = equal_bytes8(ring_ld8(ring, ring_fwd(qmo, ml + 8 * lane)), ring_ld8(ring, ring_fwd(cmo, ml + 8 * lane)));
                            if (same > mlimit - a) same = mlimit - a;
                        }
                        const unsigned long long brk = __ballot(same < 8);
                        if (brk) {
                            const uint32_t fl = (uint32_t)__ffsll((long long)brk) - 1;
                            ml += 8 * fl + wave_readlane(same, fl);
                            break;
                        }
                        ml += 512;
                    }
                    el = qm + ml; if (el > mlimit) el = mlimit;
                    e = lane == l ? el : e;
                }
                prev_end = lane == l ? wend : prev_end;
                taken |= 1ull << l;
                ntaken++;
                wend = cur = el;
                if (nseq + ntaken >= kRecsPerStrip) break;
            }
            if (taken) {
                const bool mine = (taken >> lane) & 1;
                uint32_t my_enc = 0, my_ll = 0;
                if (mine) {
                    uint32_t bk = back; if (bk > q - prev_end) bk = q - prev_end;     // lz4.c:1105-1109 over the pending literals
                    const uint32_t qs = q - bk;
                    my_ll = qs - prev_end;
                    const uint32_t mlen = e - qs;
                    MatchRec r; r.ll = my_ll; r.mo = (q - c) | ((mlen - kMinMatch) << 16);
                    recs[nseq + lanes_below(taken)] = r;
                    my_enc = enc_size(my_ll, mlen - kMinMatch);
                }
                if (nseq == 0) ll0 = wave_readlane(my_ll, (uint32_t)__ffsll((long long)taken) - 1);
                const uint32_t enc_incl = wave_incl_sum(my_enc);
                if (mine) { const uint32_t i = nseq + lanes_below(taken); ends[i] = (uint16_t)(e - cs); encp[i] = (uint16_t)(enc + enc_incl - my_enc); }
                enc += wave_readlane(enc_incl, 63);
                nseq += ntaken;
                anchor = wend;
            }
            p = cur > p + span ? (cur + smask) & ~smask : p + span;
        }
    }
    if (lane == 0) {
        strip[S_N * kCmpWaves + w] = nseq;
        strip[S_ENC * kCmpWaves + w] = enc;
        strip[S_LL0 * kCmpWaves + w] = ll0;
        strip[S_TAIL * kCmpWaves + w] = anchor < ce ? ce - anchor : 0;
        strip[S_END * kCmpWaves + w] = anchor > ce ? anchor : 0;
    }
}

// wave copy of `len` literal bytes from block position sp to dst + d: out of the ring when the
// bytes are still there, else from HBM
__device__ __forceinline__ void copy_literals(lz4amd_gdst dst, uint32_t d, lz4amd_gsrc src, const uint8_t* ring,
                                              uint32_t sp, uint32_t len, uint32_t ring_lo) {
    const uint32_t lane = lane_id();
    if (sp >= ring_lo) {
        const uint32_t o = src_ring_off(sp);
        for (uint32_t i = lane; i < len; i += 64) dst[d + i] = ring[ring_fwd(o, i)];
    } else {
        for (uint32_t i = lane; i < len; i += 64) dst[d + i] = src[sp + i];
    }
}

// ------------------------------------------------------------------------------ emit (one strip)
__device__ __forceinline__ void emit_strip(const uint8_t* ring, const MatchRec* recs, const uint32_t* strip,
                                           uint32_t w, lz4amd_gsrc src, lz4amd_gdst dst, uint32_t cs, uint32_t ring_lo) {
    const uint32_t lane = lane_id();
    const uint32_t nk = strip[S_N * kCmpWaves + w];
    const uint32_t carry = strip[S_CARRY * kCmpWaves + w];
    uint32_t ipos = cs;            // source position of the next sequence's own literals
    uint32_t opos = strip[S_OUT * kCmpWaves + w];      // dst position of the next sequence's token
    for (uint32_t base = 0; base < nk; base += 64) {
        const uint32_t i = base + lane;
        const bool have = i < nk;
        uint32_t ll = 0, mlm4 = 0, off = 0, extra = 0;
        if (have) { const MatchRec r = recs[i]; ll = r.ll; off = r.mo & 0xFFFFu; mlm4 = r.mo >> 16; }
        if (i == 0) extra = carry;                       // literals inherited from earlier strips
        const uint32_t e = have ? enc_size(ll + extra, mlm4) : 0;
        const uint32_t adv = have ? ll + mlm4 + kMinMatch : 0;
        const uint32_t e_incl = wave_incl_sum(e), a_incl = wave_incl_sum(adv);
        const uint32_t my_o = opos + e_incl - e;
        const uint32_t my_i = ipos + a_incl - adv - extra;   // source pos of my literals (carried ones included)
        uint32_t lit_dst = 0;
        const uint32_t tl = ll + extra;
        if (have) {
            lz4amd_gdst p = dst + my_o;
            const uint32_t tok_ll = tl >= 15 ? 15u : tl, tok_ml = mlm4 >= 15 ? 15u : mlm4;
            *p++ = (uint8_t)((tok_ll << 4) | tok_ml);
            if (tl >= 15) p = put_len_ext(p, tl - 15);
            lit_dst = (uint32_t)(p - dst);
            p += tl;
            p[0] = (uint8_t)off; p[1] = (uint8_t)(off >> 8); p += 2;
            if (mlm4 >= 15) p = put_len_ext(p, mlm4 - 15);
        }
        // literal runs: the short ones (nearly all) byte by byte by the lane that owns the sequence, all lanes at
        // once; the long ones one sequence at a time with the whole wave copying
        const bool shortrun = have && tl <= kShortRun;
        if (shortrun && tl) {
            if (my_i >= ring_lo) {
                // four bytes per trip (two aligned LDS dwords + a byte alignment, one dword store), then the 0-3 left over
                const uint32_t o = src_ring_off(my_i);
                const uint32_t quads = tl & ~3u;
                for (uint32_t i = 0; i < quads; i += 4) {
                    const uint32_t a = ring_fwd(o, i);
                    const uint32_t* r32 = (const uint32_t*)(ring + (a & ~3u));        // (the ring's pad covers the read past its end)
                    const uint32_t v = align_bytes(r32[1], r32[0], a & 3u);
                    __builtin_memcpy(dst + lit_dst + i, &v, 4);
                }
                for (uint32_t i = quads; i < tl; i++) dst[lit_dst + i] = ring[ring_fwd(o, i)];
            } else {
                for (uint32_t i = 0; i < tl; i++) dst[lit_dst + i] = src[my_i + i];
            }
        }
        unsigned long long longm = __ballot(have && tl > kShortRun);
        while (longm) {
            const uint32_t j = (uint32_t)__ffsll((long long)longm) - 1;
            longm &= longm - 1;
            copy_literals(dst, wave_readlane(lit_dst, j), src, ring, wave_readlane(my_i, j), wave_readlane(tl, j), ring_lo);
        }
        opos += wave_readlane(e_incl, 63);
        ipos += wave_readlane(a_incl, 63);
    }
}

// ------------------------------------------------------------------------------ offsets (one wave)
// Output offset and carried-in literals of every strip of a tile, from the strips' summaries (lane k holds
// strip k).  All strips at once: the literals carried into a strip that has sequences are the tails of the
// strips since the last one that had any (or since the previous tile); the sizes then scan to offsets.
// A strip's last match may have run past the strip's end (up to the tile's end).  The strips behind it were parsed
// at the same time, not knowing: here every strip gives up what an earlier strip's match already covers - the records
// that end inside the covered stretch go, the one that straddles its end loses its literals or the front of its match
// (same offset, the bytes are the same), or goes as well when less than a minimal match is left - and its summary is
// brought up to date.  One wave: a short serial walk over the strips settles where each one starts (a strip's own
// overrun counts only if its last match survives), then lane k puts strip k right (binary search in the record ends).
__device__ __forceinline__ void resolve_overruns(uint32_t* strip, MatchRec* recs_tile, const uint16_t* ends_tile, const uint16_t* encp_tile,
                                                 uint32_t nstrips, uint32_t t0, uint32_t strip_len, uint32_t t1, uint32_t n) {
    const uint32_t lane = lane_id();
    const bool mine = lane < nstrips;
    const uint32_t cs = t0 + lane * strip_len;
    uint32_t ce = cs + strip_len; if (ce > t1) ce = t1;
    const uint32_t nk = mine ? strip[S_N * kCmpWaves + lane] : 0, own_end = mine ? strip[S_END * kCmpWaves + lane] : 0;
    MatchRec* rk = recs_tile + lane * kRecsPerStrip;
    const uint16_t* ek = ends_tile + lane * kRecsPerStrip;
    const uint16_t* pk = encp_tile + lane * kRecsPerStrip;
    uint32_t q_last = 0;                                  // where my last match starts (strips whose last match runs over)
    if (own_end) q_last = own_end - ((rk[nk - 1].mo >> 16) + kMinMatch);
    if (!__any(own_end != 0)) {                           // nothing ran over in this tile
        if (mine) { strip[S_FIRST * kCmpWaves + lane] = 0; strip[S_P * kCmpWaves + lane] = cs; }
        return;
    }
    // ---- where does every strip start?  Behind every earlier strip's overrun: a prefix maximum - exact as long as every
    //      overrunning match survives the cut at its own strip's start (nearly always: it is long); else walk the strips.
    uint32_t P;
    {
        const uint32_t im = wave_incl_max(own_end);
        uint32_t cover = (uint32_t)__shfl_up(im, 1u); if (lane == 0) cover = 0;
        P = cover > cs ? cover : cs;
        const bool survives = own_end == 0 || P <= q_last || (own_end >= P + kMinMatch && P <= n - kMfLimit);
        if (__any(mine && !survives)) {
            cover = 0;
            for (uint32_t k = 0; k < nstrips; k++) {
                const uint32_t cs_k = t0 + k * strip_len, Pk = cover > cs_k ? cover : cs_k;
                if (lane == k) P = Pk;
                const uint32_t e_k = wave_readlane(own_end, k), q_k = wave_readlane(q_last, k);
                const bool sv = e_k != 0 && (Pk <= q_k || (e_k >= Pk + kMinMatch && Pk <= n - kMfLimit));
                if (sv && e_k > cover) cover = e_k;
            }
        }
    }
    // ---- put my strip right
    if (mine) {
        uint32_t first = 0, nk2 = nk, enc = strip[S_ENC * kCmpWaves + lane], ll0 = strip[S_LL0 * kCmpWaves + lane], tail = strip[S_TAIL * kCmpWaves + lane];
        if (P > cs) {
            uint32_t lo = 0, hi = nk;                         // first record whose match ends behind P
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cs + ek[mid] > P) hi = mid; else lo = mid + 1; }
            uint32_t f = lo;
            first = nk; nk2 = 0; enc = 0; ll0 = 0; tail = ce > P ? ce - P : 0;
            if (f < nk) {
                MatchRec r = rk[f];
                const uint32_t ef = cs + ek[f], qf = ef - ((r.mo >> 16) + kMinMatch);
                const uint32_t old_enc = enc_size(r.ll, r.mo >> 16), enc_total = strip[S_ENC * kCmpWaves + lane];
                bool dropped = false;
                if (P > qf) {
                    const uint32_t left = ef - P;
                    dropped = left < kMinMatch || P > n - kMfLimit;         // a match starts at least 12 bytes before the block's end (lz4.c:1030)
                    if (!dropped) { r.ll = 0; r.mo = (r.mo & 0xFFFFu) | ((left - kMinMatch) << 16); }
                } else r.ll = qf - P;
                if (!dropped) {
                    rk[f] = r;
                    first = f; nk2 = nk - f; ll0 = r.ll;
                    enc = enc_total - pk[f] - old_enc + enc_size(r.ll, r.mo >> 16);
                } else if (f + 1 < nk) {
                    // its bytes [P, ef) are literals of the next record
                    MatchRec r2 = rk[f + 1];
                    const uint32_t old2 = enc_size(r2.ll, r2.mo >> 16);
                    r2.ll += ef - P;
                    rk[f + 1] = r2;
                    first = f + 1; nk2 = nk - f - 1; ll0 = r2.ll;
                    enc = enc_total - pk[f + 1] - old2 + enc_size(r2.ll, r2.mo >> 16);
                }
                if (nk2) { const uint32_t e_last = cs + ek[nk - 1]; tail = e_last < ce ? ce - e_last : 0; }
            }
        }
        strip[S_N * kCmpWaves + lane] = nk2; strip[S_ENC * kCmpWaves + lane] = enc; strip[S_LL0 * kCmpWaves + lane] = ll0;
        strip[S_TAIL * kCmpWaves + lane] = tail; strip[S_FIRST * kCmpWaves + lane] = first; strip[S_P * kCmpWaves + lane] = P;
    }
}

struct StripTotals { uint32_t out, carry, fail; };
__device__ __forceinline__ StripTotals strip_offsets(uint32_t* strip, uint32_t nstrips, uint32_t out0, uint32_t carry0,
                                                     uint32_t fail, uint32_t cap) {
    const uint32_t lane = lane_id();
    const bool mine = lane < nstrips;
    const uint32_t nk = mine ? strip[S_N * kCmpWaves + lane] : 0, en = mine ? strip[S_ENC * kCmpWaves + lane] : 0;
    const uint32_t l0 = mine ? strip[S_LL0 * kCmpWaves + lane] : 0, tl = mine ? strip[S_TAIL * kCmpWaves + lane] : 0;
    const unsigned long long ne = __ballot(mine && nk != 0);                 // strips with sequences
    const uint32_t t_incl = wave_incl_sum(tl), t_excl = t_incl - tl;
    const unsigned long long before = ne & ((1ull << lane) - 1);
    const uint32_t prev = before ? 63u - (uint32_t)__clzll(before) : 0u;       // the last such strip before me
    const uint32_t t_prev = (uint32_t)__shfl((int)t_excl, (int)prev);
    const uint32_t my_carry = before ? t_excl - t_prev : carry0 + t_excl;
    const uint32_t sz = nk ? en + my_carry + lit_hdr_ext(l0 + my_carry) - lit_hdr_ext(l0) : 0u;
    const uint32_t o_incl = wave_incl_sum(sz);
    // a strip is only written if it fits (the block then fails as a whole, lz4.c:1116)
    if (__ballot(nk != 0 && (uint64_t)out0 + o_incl > cap)) fail = 1;
    const uint32_t total = wave_readlane(o_incl, 63), t_total = wave_readlane(t_incl, 63);
    const uint32_t last = ne ? 63u - (uint32_t)__clzll(ne) : 0u;
    if (mine) { strip[S_OUT * kCmpWaves + lane] = out0 + o_incl - sz; strip[S_CARRY * kCmpWaves + lane] = my_carry; }
    StripTotals r;
    r.out = fail ? out0 : out0 + total;
    r.carry = ne ? t_total - wave_readlane(t_excl, last) : carry0 + t_total;
    r.fail = fail;
    return r;
}

// a control word every lane of the wave agrees on
__device__ __forceinline__ uint32_t uload_cm(const uint32_t* w) { return __builtin_amdgcn_readfirstlane(lds_load_acquire(w)); }

// ------------------------------------------------------------------------------ one block
__device__ __forceinline__ void compress_one_block(const CompBatch& P, uint32_t b, char* smem) {
    const uint32_t tid = threadIdx.x, w = wave_id();
    uint32_t* misc = (uint32_t*)(smem + kCOffMisc);
    uint32_t* strip = (uint32_t*)(smem + kCOffStrip);
    uint32_t* tab = (uint32_t*)(smem + kCOffTab);
    MatchRec* recs = (MatchRec*)(smem + kCOffRecs) + w * kRecsPerStrip;      // + parity * kCmpWaves * kRecsPerStrip
    uint16_t* ends = (uint16_t*)(smem + kCOffEnds) + w * kRecsPerStrip;
    uint16_t* encp = (uint16_t*)(smem + kCOffEncp) + w * kRecsPerStrip;
    uint8_t* ring = (uint8_t*)(smem + kCOffRing);

    // history (linked blocks, lz4io.c:741-744 / LZ4_compress_fast_continue in prefix mode lz4.c:1707): the
    // `pre` bytes right before the block are parsed into the table but not emitted.  Whole tiles only.
    uint32_t pre = P.prefix ? (uint32_t)P.prefix[b] : 0u;
    if (pre > kMaxDistance + 1) pre = kMaxDistance + 1;
    pre &= ~(kTileMax - 1);
    const lz4amd_gsrc src = LZ4AMD_TO_GSRC(P.src[b]) - pre;          // position 0 = start of the history
    const lz4amd_gdst dst = LZ4AMD_TO_GDST(P.dst[b]);
    const int32_t n_i = P.src_size[b];
    const int32_t cap_i = P.dst_cap[b];
    if (n_i < 0 || (uint32_t)n_i > kMaxInput || cap_i <= 0 || P.dst[b] == nullptr || (P.src[b] == nullptr && n_i != 0)) {
        if (tid == 0) P.result[b] = 0;                               // lz4.c:1360
        return;
    }
    if (n_i == 0) { if (tid == 0) { dst[0] = 0; P.result[b] = 1; } return; }      // lz4.c:1361-1371
    const uint32_t n = (uint32_t)n_i + pre, cap = (uint32_t)cap_i;
    const bool small = n < kSmallBlockLimit;

    for (uint32_t i = tid; i < (1u << kHashBits); i += kCmpThreads) tab[i] = 0;
    if (tid == 0) { misc[CM_OUT] = 0; misc[CM_CARRY] = 0; misc[CM_FAIL] = 0; misc[CM_READY] = 0; }
    // first tile straight into the ring; later tiles are prefetched one tile ahead
    uint32_t t0 = 0, tile_len, strip_len;
    tile_geometry(pre ? kTileMax * 4 : 0, small, tile_len, strip_len);
    uint32_t loaded = 0;                                  // ring holds [.., loaded)
    uint64_t* prof = P.prof ? P.prof + (uint64_t)blockIdx.x * 8 : nullptr;
    uint64_t tp[5] = {0, 0, 0, 0, 0}, tq = 0;
    if (prof) tq = clock_ticks();
    {
        uint32_t hi = tile_len + 16; if (hi > n) hi = n;
        for (uint32_t Pp = 16 * tid; Pp < hi; Pp += 16 * kCmpThreads) ring_commit16(ring, Pp, load_src16(src, n, Pp));
        loaded = (hi + 15) & ~15u;
    }
    // Two barriers per tile.  Interval A: wave 0 first settles tile k-1 (overrunning matches, then the strips' sizes
    // into output offsets: ~3.5 K cycles of one wave's dependent work, which used to sit between the barriers with
    // fifteen waves waiting) and says so in CM_READY; every wave parses its strip of tile k, then - once CM_READY
    // covers tile k-1, which it long does by then - writes out its strip of tile k-1.  Interval B: everybody inserts
    // tile k into the table.  Records and strip summaries are double buffered for that.
    uint32_t par = 0;                                       // buffer parity of tile k
    uint32_t prev_t0 = 0, prev_t1 = 0, prev_strip_len = 0, prev_nstrips = 0;      // tile k-1, still to be emitted
    uint32_t tiles_parsed = 0;                              // tiles whose strips were matched so far (CM_READY counts up to it)
    while (t0 < n) {
        uint32_t t1 = t0 + tile_len; if (t1 > n || t1 < t0) t1 = n;
        uint32_t* strip_k = strip + par * kStripFields * kCmpWaves;
        uint32_t* strip_p = strip + (par ^ 1) * kStripFields * kCmpWaves;
        MatchRec* recs_k = recs + par * kCmpWaves * kRecsPerStrip;
        MatchRec* recs_p = recs + (par ^ 1) * kCmpWaves * kRecsPerStrip;
        // -- prefetch: the next tile's bytes (one 16-byte granule per thread, committed after the parse)
        uint32_t nt_len, nt_strip;
        tile_geometry(pre ? kTileMax * 4 : t1, small, nt_len, nt_strip);
        uint32_t pf_hi = loaded + nt_len; if (pf_hi > n || pf_hi < loaded) pf_hi = n;     // stays 16 bytes ahead of the tile
        const uint32_t Pp = loaded + 16 * tid;
        U32x4 pf; pf[0] = pf[1] = pf[2] = pf[3] = 0;
        if (Pp < pf_hi) pf = load_src16(src, n, Pp);           // nt_len <= 16 * kCmpThreads
        __syncthreads();                                       // ring, table and tile k-1's offsets ready
        if (prof) { const uint64_t t = clock_ticks(); tp[0] += t - tq; tq = t; }
        // -- A0: wave 0 settles tile k-1
        if (w == 0 && prev_nstrips) {
            resolve_overruns(strip_p, recs_p - w * kRecsPerStrip, ends + (par ^ 1) * kCmpWaves * kRecsPerStrip - w * kRecsPerStrip,
                             encp + (par ^ 1) * kCmpWaves * kRecsPerStrip - w * kRecsPerStrip, prev_nstrips, prev_t0, prev_strip_len, prev_t1, n);
            wave_lds_fence();
            const StripTotals t = strip_offsets(strip_p, prev_nstrips, misc[CM_OUT], misc[CM_CARRY], misc[CM_FAIL], cap);
            if (lane_id() == 0) { misc[CM_OUT] = t.out; misc[CM_CARRY] = t.carry; misc[CM_FAIL] = t.fail; }
            wave_lds_fence();
            if (lane_id() == 0) lds_store_release(&misc[CM_READY], tiles_parsed);
        }
        // -- A1: match, one wave per strip (tiles of the history are only inserted into the table)
        const bool parse = t0 >= pre;
        const uint32_t nstrips = parse ? (t1 - t0 + strip_len - 1) / strip_len : 0;
        if (w < nstrips) {
            const uint32_t cs = t0 + w * strip_len;
            uint32_t ce = cs + strip_len; if (ce > t1) ce = t1;
            match_strip(ring, tab, recs_k, ends + par * kCmpWaves * kRecsPerStrip, encp + par * kCmpWaves * kRecsPerStrip, strip_k, w, n, cs, ce, t1);
        }
        if (prof) { const uint64_t t = clock_ticks(); tp[1] += t - tq; tq = t; }
        // -- A2: emit tile k-1
        const uint32_t ring_lo = loaded > kSrcRing ? loaded - kSrcRing : 0;
        if (w < prev_nstrips) {
            while (uload_cm(&misc[CM_READY]) < tiles_parsed) spin_pause();
            if (!misc[CM_FAIL] && strip_p[S_N * kCmpWaves + w])
                emit_strip(ring, recs_p + strip_p[S_FIRST * kCmpWaves + w], strip_p, w, src, dst, strip_p[S_P * kCmpWaves + w], ring_lo);
        }
        if (prof) { const uint64_t t = clock_ticks(); tp[4] += t - tq; tq = t; }
        __syncthreads();
        if (prof) { const uint64_t t = clock_ticks(); tp[2] += t - tq; tq = t; }
        // -- B: everybody inserts the tile into the table (positions that may start a match):
        //    8 consecutive positions per thread, hashed out of four aligned dwords
        if (n >= kMfLimit + 1) {
            const uint32_t last_q = n - kMfLimit;
            const uint32_t q0 = t0 + 8 * tid;                       // t0 is a multiple of 1024; tiles are at most 8 * kCmpThreads long
            if (q0 < t1 && q0 <= last_q) {
                const uint32_t o = src_ring_off(q0);                // multiple of 8: o + 16 <= ring + pad
                const uint32_t* a = (const uint32_t*)(ring + o);
                uint32_t dw[4];
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) dw[i] = a[i];
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) {
                    const uint32_t q = q0 + i;
                    const uint32_t lo = align_bytes(dw[i / 4 + 1], dw[i / 4], i & 3), hi = align_bytes(dw[i / 4 + 2], dw[i / 4 + 1], i & 3);
                    if (q < t1 && q <= last_q) atomicMax(&tab[hash_pos32(lo, hi, small)], q);
                }
            }
        }
        // the prefetched granules go into ring slots that hold bytes more than a window + two tiles old
        if (Pp < pf_hi) ring_commit16(ring, Pp, pf);
        if (pf_hi > loaded) loaded = (pf_hi + 15) & ~15u;
        if (prof) { const uint64_t t = clock_ticks(); tp[3] += t - tq; tq = t; }
        prev_t0 = t0; prev_t1 = t1; prev_strip_len = strip_len; prev_nstrips = nstrips; par ^= 1;
        if (nstrips) tiles_parsed++;
        t0 = t1; tile_len = nt_len; strip_len = nt_strip;
    }
    __syncthreads();
    // -- the last tile's sequences (settled by wave 0 first)
    {
        uint32_t* strip_p = strip + (par ^ 1) * kStripFields * kCmpWaves;
        MatchRec* recs_p = recs + (par ^ 1) * kCmpWaves * kRecsPerStrip;
        if (w == 0 && prev_nstrips) {
            resolve_overruns(strip_p, recs_p - w * kRecsPerStrip, ends + (par ^ 1) * kCmpWaves * kRecsPerStrip - w * kRecsPerStrip,
                             encp + (par ^ 1) * kCmpWaves * kRecsPerStrip - w * kRecsPerStrip, prev_nstrips, prev_t0, prev_strip_len, prev_t1, n);
            wave_lds_fence();
            const StripTotals t = strip_offsets(strip_p, prev_nstrips, misc[CM_OUT], misc[CM_CARRY], misc[CM_FAIL], cap);
            if (lane_id() == 0) { misc[CM_OUT] = t.out; misc[CM_CARRY] = t.carry; misc[CM_FAIL] = t.fail; }
        }
        __syncthreads();
        const uint32_t ring_lo = loaded > kSrcRing ? loaded - kSrcRing : 0;
        if (w < prev_nstrips && !misc[CM_FAIL] && strip_p[S_N * kCmpWaves + w])
            emit_strip(ring, recs_p + strip_p[S_FIRST * kCmpWaves + w], strip_p, w, src, dst, strip_p[S_P * kCmpWaves + w], ring_lo);
    }
    __syncthreads();
    if (prof) {
        // developer aid: match + emit time of every wave (spread between the strips of a tile)
        if (lane_id() == 0) misc[8 + w] = (uint32_t)((tp[1] + tp[4]) >> 4);
        __syncthreads();
        if (tid == 0) {
            uint64_t mx = 0, mn = ~0ull, sm = 0;
            for (uint32_t i = 0; i < kCmpWaves; i++) { const uint64_t v = (uint64_t)misc[8 + i] << 4; mx = v > mx ? v : mx; mn = v < mn ? v : mn; sm += v; }
            prof[0] = tp[0]; prof[1] = tp[1]; prof[2] = tp[2]; prof[3] = tp[3]; prof[4] = tp[4]; prof[5] = mx; prof[6] = mn; prof[7] = sm / kCmpWaves;
        }
    }
    // -- final literal run (lz4.c:1302-1329)
    const uint32_t out = misc[CM_OUT], run = misc[CM_CARRY];
    const uint64_t total = (uint64_t)out + 1 + lit_hdr_ext(run) + run;
    if (misc[CM_FAIL] || total > cap) { if (tid == 0) P.result[b] = 0; return; }
    const uint32_t lit_dst = out + 1 + lit_hdr_ext(run);
    if (tid == 0) {
        lz4amd_gdst p = dst + out;
        if (run >= 15) { *p++ = 0xF0; put_len_ext(p, run - 15); }
        else *p++ = (uint8_t)(run << 4);
        P.result[b] = (int32_t)total;
    }
    {
        const uint32_t sp = n - run;
        const uint32_t full = run & ~15u;
        for (uint32_t i = 16 * tid; i < full; i += 16 * kCmpThreads) st_global16(dst + lit_dst + i, ld_global16(src + sp + i));
        for (uint32_t i = full + tid; i < run; i += kCmpThreads) dst[lit_dst + i] = src[sp + i];
    }
}

// Workgroups pull blocks from a device-wide ticket counter (load balance for ragged batches).
__device__ __forceinline__ void compress_batch_body(const CompBatch& P) {
    LZ4AMD_DYN_LDS(smem);
    uint32_t* misc = (uint32_t*)(smem + kCOffMisc);
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) misc[CM_BLOCK] = take_ticket(P.ticket);
        __syncthreads();
        const uint32_t b = misc[CM_BLOCK];
        if (b >= P.n_blocks) break;
        compress_one_block(P, b, smem);
    }
}

} // namespace lz4amd
