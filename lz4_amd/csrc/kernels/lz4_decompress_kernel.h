// lz4_decompress_kernel.h -- batched LZ4 block decompression for gfx950 (MI355X), streaming design.
//
// Replaces, for a whole batch of independent blocks resident in HBM, what the reference does per
// block in LZ4_decompress_safe (lib/lz4.c:2451 -> LZ4_decompress_generic lz4.c:2023-2445; length
// fields: read_variable_length lz4.c:1979-2014; end-of-block rules lz4.c:2276-2330, 2421-2429).
// Accepts ANY legal LZ4 block, rejects what the reference's safe loop rejects, never reads outside
// src[0,csize) nor writes outside dst[0,cap).  Not a port: the reference decoder is one serial
// token chain per block.  Here ONE 1024-thread workgroup (16 waves, one CU, all of its 160 KB LDS)
// decodes a block as a three-role pipeline; the roles talk through LDS rings and counters only -
// no workgroup barrier between the first and the last byte of a block, no scratch in HBM:
//
//   LOADER (wave 15)   streams the compressed block into a 48 KB LDS ring for the parsers
//       (coalesced 16-byte loads, the next 4 KB in flight while the last is written).
//
//   PARSERS (4 waves)  turn the serial token chain into sequence records {output position,
//       literal source, literal length, offset} in a 1024-row LDS ring.  The stream is cut in
//       TILES of 8 KB on a fixed grid, tile i belongs to parser wave i mod 4; a tile is cut in
//       64 segments of 128 B, one per lane.
//         P1  every lane walks the chain of its segment from the segment's first byte, marking
//             the token positions it visits in a bitmap.  A wrong start walks over literal bytes
//             misread as tokens (~6.5 B per step) and merges with the true chain after a few
//             hundred bytes (LZ4 chains self-synchronise).
//         P2  every lane walks on from its exit (the "bridge") until it steps on a position a
//             later lane marked, or the tile ends.
//         (P1 + P2 need nothing from the earlier tiles: the four owners walk side by side.)
//         P3  once the previous tile's owner has handed over the ENTRY - the first true token at
//             or after the tile's start - a short walk from it finds the first marked position;
//             the lane that marked it is true from there on, and where a true lane's bridge
//             merged into lane k's marks, lane k is true from there on (<= 64 hops).  A bridge
//             that never merged ends the stitch early at its last (true) position - never a
//             wrong answer: the owner walks the rest of the tile again from that position.
//         P4  marks before a lane's merge point and of skipped lanes are dropped, the bridges of
//             the true lanes are added: the bitmap now holds exactly the true tokens.
//         P5  64 tokens at a time, a lane decodes one sequence completely (both length fields,
//             offset) and checks that it starts where its predecessor ended - the decoder, not
//             the walk, is the authority on the chain; wave scans place the sequences relative
//             to the tile.  Then, when it is the owner's TURN (output positions are a running
//             sum over all earlier sequences), the reference's output-side rules are applied
//             and the records are published.
//       Tokens whose fields leave the tile (+1 KB look-ahead), length fields longer than 32
//       bytes and the block's last sequence go through a wave-cooperative SLOW PATH that takes
//       one token at a time (any length, records split at 8 KB, flow control inside).
//
//   COPY (waves 0-10)  output-stationary in 1 KB REGIONS, wave w owns regions w, w+11, ... .
//       A region is composed in its slot of an 80 KB LDS ring that always holds the 64 KB LZ4
//       window, from PIECES (the literal run or the match of a record, cut at 16-byte chunk
//       borders) in two lane-uniform rounds: round A, lane = chunk, writes the piece that covers
//       the chunk's first byte; round B, lane = piece, ORs in the head of every piece that starts
//       inside a chunk.  Literals are unaligned 16-byte loads from the block itself (the L2 still
//       holds what the loader fetched), matches unaligned 16-byte reads from the output ring (a match that overlaps itself reads any earlier period - the
//       farthest the window holds - so long runs do not serialise).  Sources still in flight on
//       another wave are waited for through per-chunk done bits; finished regions go to HBM
//       with one 16-byte store per lane (1 KB contiguous per wave).
//
// HBM traffic per block: compressed bytes read once (the literal loads hit the L2), output written
// once.  No MFMA: byte moves.
#pragma once
#include "lz4_common.h"
#include "../lz4amd_params.h"

#ifdef LZ4AMD_TRACE
#define DTRACE(...) do { if (lane_id() == 0) { fprintf(stderr, "[w%u] ", wave_id()); fprintf(stderr, __VA_ARGS__); } } while (0)
#else
#define DTRACE(...) do {} while (0)
#endif

namespace lz4amd {

using DecBatch = ::lz4amd_dec_params;     // argument block (lz4amd_params.h)

struct alignas(16) SeqRec { uint32_t outpos, litpos, ll, off; };
struct alignas(16) DoneEnt { uint64_t mask; uint32_t tag, pad; };

enum : uint32_t {
    kDecThreads = 1024,
    kDecWaves = kDecThreads / 64,
    kParseWaves = 4,                            // tile owners
    kLoadWave = kDecWaves - 1,
    kCopyWaves = kDecWaves - 1 - kParseWaves,   // waves 0 .. kCopyWaves-1
    kChunk = 16,                                // output bytes composed at a time
    kRegionShift = 10,
    kRegion = 1u << kRegionShift,               // 64 chunks
    kSlots = 80,                                // output ring slots (regions): 64 KB window + regions in flight
    kRingBytes = kSlots * kRegion,
    kRingPad = 32,                              // mirror of the first bytes: reads never wrap
    kMaxLead = kSlots - 64 - 1,                 // a wave may lead the first unfinished region by this many
    kCrBytes = 48u << 10,                       // compressed ring (direct mapped: position mod 48 K), read by the parsers only
    kCrPad = 32,
    kLoadBatch = 4096,                          // bytes the loader moves per step
    kRecCap = 1024,                             // sequence-record ring
    kRecMask = kRecCap - 1,
    kIdxRing = 512,                             // first record of a region, per region (ring)
    kIdxMask = kIdxRing - 1,
    kOutAhead = 480,                            // records are published at most this many regions ahead of the copy
    kSegShift = 7,
    kSeg = 1u << kSegShift,                     // parser segment (bytes of the stream per lane)
    kTile = 64 * kSeg,
    kLook = 1024,                               // fields of a tile's tokens may reach this far past the tile
    kBridgeTrips = 40,                          // lockstep trips of the bridge walk
    kEntrySteps = 64,                           // steps of the serial walk from a tile's entry to the first mark
    kTokRound = 512,                            // tokens listed per round
    kExtMax = 32,                               // longer length fields take the slow path
    kSlowSpan = 8192,                           // slow-path records cover at most this many output bytes
    kMaxTrips = 10,                             // round-B trips per region (32 records each)
    kBias = 65536,                              // output positions are biased: [kBias - prefix, kBias) is the history before dst
    kFirstRegion = kBias >> kRegionShift,
    kNone = 0xFFFFFFFFu,
};
static_assert(kParseWaves * kTile + kLook + kLoadBatch <= kCrBytes, "the tiles in flight must fit the compressed ring");

// LDS carve-up (bytes)
enum : uint32_t {
    kOffMisc = 0,                                            // u32[64] control words
    kOffFin = kOffMisc + 64 * 4,                             // u32[16] regions completed per copy wave
    kOffBits = kOffFin + 16 * 4,                             // DoneEnt[kSlots]
    kOffIdx = kOffBits + kSlots * 16,                        // u16[kIdxRing]
    kOffPend = kOffIdx + kIdxRing * 2,                       // u64[kCopyWaves][kMaxTrips + 2] pending masks of round B
    kOffPar = kOffPend + kCopyWaves * (kMaxTrips + 2) * 8,   // per parser wave: token bitmap of its tile, token list
    kParBytes = kTile / 8 + kTokRound * 2,
    kOffRecs = (kOffPar + kParseWaves * kParBytes + 15) & ~15u,   // SeqRec[kRecCap]
    kOffCr = kOffRecs + kRecCap * 16,                        // compressed ring + pad
    kOffRing = kOffCr + kCrBytes + kCrPad,                   // output ring + pad
    kDecLdsBytes = kOffRing + kRingBytes + kRingPad,
};
static_assert(kDecLdsBytes <= 160u * 1024u, "LDS budget");
static_assert((kOffRecs % 16) == 0 && (kOffCr % 16) == 0 && (kOffRing % 16) == 0 && (kOffBits % 16) == 0 && (kOffPend % 8) == 0 && (kOffPar % 4) == 0, "LDS alignment");

enum : uint32_t { M_BLOCK = 0, M_ERR, M_ABORT, M_FIN, M_CHI, M_CLO, M_EMIT, M_HEAD, M_ENT_SEQ, M_ENT_POS, M_TURN };

// (the round-1 decoder kept a record table in HBM; this one needs no scratch)
__host__ __device__ inline uint64_t dec_scratch_bytes(uint32_t) { return 256; }

// ------------------------------------------------------------------------------ small helpers
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }
// position in the compressed ring
__device__ __forceinline__ uint32_t mod_cr(uint32_t x) {            // x mod 48 K, x < 2^31
    return x - (uint32_t)(((uint64_t)(x >> 14) * 0xAAAAAAABull) >> 33) * kCrBytes;
}
__device__ __forceinline__ uint32_t cr_fold(uint32_t a) { return umin32(a, a - kCrBytes); }        // [0, 2*48K) -> [0, 48K)
__device__ __forceinline__ uint32_t ring_fold(uint32_t a) { return umin32(a, a - kRingBytes); }
// a control word every lane of the wave agrees on
__device__ __forceinline__ uint32_t uload(const uint32_t* w) { return __builtin_amdgcn_readfirstlane(lds_load_acquire(w)); }

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
// dword k of the 16-byte mask that selects bytes [0, n), n in 0..16
__device__ __forceinline__ uint32_t low_bytes_mask(uint32_t n, uint32_t k) {
    const int32_t r = (int32_t)n - 4 * (int32_t)k;
    return r >= 4 ? 0xFFFFFFFFu : (r <= 0 ? 0u : ((1u << (8 * r)) - 1u));
}
// v restricted to bytes [lo, hi)
__device__ __forceinline__ U32x4 keep_bytes(const U32x4& v, uint32_t lo, uint32_t hi) {
    U32x4 r;
    r[0] = v[0] & low_bytes_mask(hi, 0) & ~low_bytes_mask(lo, 0);
    r[1] = v[1] & low_bytes_mask(hi, 1) & ~low_bytes_mask(lo, 1);
    r[2] = v[2] & low_bytes_mask(hi, 2) & ~low_bytes_mask(lo, 2);
    r[3] = v[3] & low_bytes_mask(hi, 3) & ~low_bytes_mask(lo, 3);
    return r;
}
// 16 bytes starting at ANY byte a of an LDS array of dwords (the arrays are padded: a + 20 is in range)
__device__ __forceinline__ U32x4 lds_read16_at(const uint8_t* base, uint32_t a) {
    const uint32_t* r32 = (const uint32_t*)(base + (a & ~3u));
    const uint32_t sh = a & 3u;
    const uint32_t d0 = r32[0], d1 = r32[1], d2 = r32[2], d3 = r32[3], d4 = r32[4];
    U32x4 v;
    v[0] = align_bytes(d1, d0, sh); v[1] = align_bytes(d2, d1, sh);
    v[2] = align_bytes(d3, d2, sh); v[3] = align_bytes(d4, d3, sh);
    return v;
}
// 16 bytes of the compressed stream at position P (tail of the block zero padded; never reads past csize)
__device__ __forceinline__ U32x4 load_granule(lz4amd_gsrc src, uint32_t csize, uint32_t P) {
    if (P + 16 <= csize) return ld_global16(src + P);
    U32x4 v; v[0] = v[1] = v[2] = v[3] = 0;
#pragma nounroll
    for (uint32_t i = 0; i < 16 && P + i < csize; i++) chunk_set_byte(v, i, (uint32_t)src[P + i]);
    return v;
}

// First region some copy wave has not completed: every region below it is final.
__device__ __forceinline__ uint32_t first_open_region(const char* smem) {
    const uint32_t* fin = (const uint32_t*)(smem + kOffFin);
    const uint32_t l16 = lane_id() & 15u;
    const uint32_t f = lds_load_acquire(&fin[l16]);
    const uint32_t r = l16 < kCopyWaves ? kFirstRegion + l16 + kCopyWaves * f : kNone;
    return __builtin_amdgcn_readfirstlane(row16_min_u32(r));
}
__device__ __forceinline__ bool block_over(const char* smem) {        // finished or failed: nothing left to wait for
    const uint32_t* misc = (const uint32_t*)(smem + kOffMisc);
    return (uload(&misc[M_FIN]) | uload(&misc[M_ABORT])) != 0;
}

// ------------------------------------------------------------------------------ LOADER
__device__ __forceinline__ void loader_role(lz4amd_gsrc src, uint32_t csize, char* smem) {
    uint8_t* cr = (uint8_t*)(smem + kOffCr);
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    const uint32_t lane = lane_id();
    U32x4 cur[4], nxt[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) { const uint32_t P = 16 * (lane + 64 * i); if (P < csize) cur[i] = load_granule(src, csize, P); }
    uint32_t L = 0;
    while (L < csize) {
        // bytes [L, L + batch) may be written once no parser needs the bytes 48 K below them
        for (;;) {
            const uint32_t clo = uload(&misc[M_CLO]);
            if (L + kLoadBatch <= clo + kCrBytes) break;
            if (block_over(smem)) return;
            spin_pause_long();
        }
        const uint32_t nL = L + kLoadBatch;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) { const uint32_t P = nL + 16 * (lane + 64 * i); if (P < csize) nxt[i] = load_granule(src, csize, P); }
        const uint32_t a0 = mod_cr(L);
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            const uint32_t o = 16 * (lane + 64 * i);
            if (L + o < csize) {
                const uint32_t a = cr_fold(a0 + o);
                *(U32x4*)(cr + a) = cur[i];
                if (a < kCrPad) *(U32x4*)(cr + kCrBytes + a) = cur[i];
            }
        }
        wave_lds_fence();
        if (lane == 0) lds_store_release(&misc[M_CHI], nL < csize ? nL : csize);
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) cur[i] = nxt[i];
        L = nL;
    }
}

// ------------------------------------------------------------------------------ PARSER
// The stream is cut in TILES of kTile bytes on a fixed grid; tile i belongs to parser wave i mod kParseWaves.
// What a tile's owner can do before it knows where the true chain enters the tile (the speculative walk) runs
// in parallel with the other owners; two short hand-offs are serial from tile to tile: the ENTRY (the first
// true token at or after the tile's start, known once the previous tile is stitched) and the TURN to publish
// records (output positions are a running sum over all earlier sequences).
struct ParserS {
    uint32_t csize, capB, low;      // capB = capacity + kBias; low = first output position that exists (kBias - prefix)
    uint32_t e;                     // slow path: token to decode / next token after it
    uint32_t obase, head;           // output position of the next sequence / records published (valid while holding the turn)
    uint32_t ppos;                  // slow path: lowest compressed position still needed
    uint32_t g, tail;               // first open region / first record still in use (last refresh)
    uint64_t t_wait, t_walk, t_stitch, t_decode, t_turn, t_slow;   // developer profile (cycles)
    uint32_t n_trips;
};

// Look at the copy waves' progress: first open region and first record still needed.
__device__ __forceinline__ void parser_refresh(ParserS& S, char* smem) {
    const uint16_t* idx = (const uint16_t*)(smem + kOffIdx);
    const uint32_t g = first_open_region(smem);
    uint32_t tail = S.head;
    if ((g << kRegionShift) < S.obase) {                     // region g is covered by published records
        const uint32_t t16 = idx[g & kIdxMask];
        tail = S.head - ((S.head - t16) & 0xFFFFu);
    }
    S.g = g; S.tail = tail;
}
__device__ __forceinline__ void set_clo(char* smem, uint32_t pos) {            // (turn holder only; never moves back)
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    if (lane_id() == 0 && pos > misc[M_CLO]) lds_store_release(&misc[M_CLO], pos);
}
// wait until the stream is resident up to `need`; false: the block is over
__device__ __forceinline__ bool parser_wait_data(ParserS& S, char* smem, uint32_t need) {
    const uint32_t* misc = (const uint32_t*)(smem + kOffMisc);
    if (need > S.csize) need = S.csize;
    if (uload(&misc[M_CHI]) >= need) return true;
    const uint64_t t0 = clock_ticks();
    bool ok = true;
    for (;;) {
        if (uload(&misc[M_CHI]) >= need) break;
        if (block_over(smem)) { ok = false; break; }
        spin_pause();
    }
    S.t_wait += clock_ticks() - t0;
    return ok;
}
__device__ __forceinline__ void parser_fail(char* smem, uint32_t pos) {
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    if (lane_id() == 0) { lds_store_relaxed(&misc[M_ERR], pos); lds_store_release(&misc[M_ABORT], 1u); }
}

// the compressed ring as a tile's walkers see it
struct TileView { const uint8_t* cr; uint32_t t0, crT, tlim, csize; };
__device__ __forceinline__ uint32_t tv_byte(const TileView& V, uint32_t p) { return (uint32_t)V.cr[cr_fold(V.crT + (p - V.t0))]; }

struct TokInfo { uint32_t ll, q, off, ml, nx, st; };      // st: 0 decoded, 1 not inside the tile's reach (slow path), 2 malformed
// Decode the sequence whose token is at p (p < tlim).  The walkers (FULL = false) only need nx; the
// same function with FULL = true is the authority on the fields and on the reference's input-side
// rules (read_variable_length, lz4.c:1979-2014).
template <bool FULL>
__device__ __forceinline__ TokInfo tok_decode(const TileView& V, uint32_t p) {
    TokInfo r; r.ll = 0; r.q = 0; r.off = 0; r.ml = 0; r.nx = 0; r.st = 1;
    const uint32_t b = tv_byte(V, p);
    uint32_t ll = b >> 4, q = p + 1;
    if (ll == 15) {
        uint32_t n = 0, x;
        do {
            if (q >= V.tlim || n >= kExtMax) return r;
            if (FULL && q + 15 >= V.csize) { r.st = 2; return r; }
            x = tv_byte(V, q); q++; n++; ll += x;
        } while (x == 255);
    }
    r.ll = ll; r.q = q;
    if (V.csize - q < ll + 8) return r;                  // the block's last sequence (or a malformed one): slow path
    const uint32_t m = q + ll;
    if (m + 2 > V.tlim) return r;
    uint32_t nx = m + 2, ml = b & 15;
    if (ml == 15) {
        uint32_t n = 0, x;
        do {
            if (nx >= V.tlim || n >= kExtMax) return r;
            x = tv_byte(V, nx); nx++; n++; ml += x;
            if (FULL && nx + 4 > V.csize) { r.st = 2; return r; }
        } while (x == 255);
    }
    if (FULL) r.off = tv_byte(V, m) | (tv_byte(V, m + 1) << 8);
    r.ml = ml + kMinMatch; r.nx = nx; r.st = 0;
    return r;
}

// 8 stream bytes at position p of the tile view (any alignment; the ring is padded, bytes past tlim are garbage, not faults)
__device__ __forceinline__ uint64_t tv_read8(const TileView& V, uint32_t p) {
    const uint32_t a = cr_fold(V.crT + (p - V.t0));
    const uint32_t* w = (const uint32_t*)(V.cr + (a & ~3u));
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], sh = a & 3u;
    return (uint64_t)align_bytes(w1, w0, sh) | ((uint64_t)align_bytes(w2, w1, sh) << 32);
}
// tok_decode<true> for the common shapes - both length fields at most 5 bytes, everything well inside the tile's reach
// and the block - with two 8-byte reads; any other token goes through tok_decode<true> itself (same answers).
__device__ __forceinline__ TokInfo tok_decode_fast(const TileView& V, uint32_t p) {
    TokInfo r; r.st = 0;
    bool easy = p + 8 <= V.tlim;
    const uint64_t w = tv_read8(V, p);
    const uint32_t t = (uint32_t)w & 0xFFu, lnib = t >> 4, mnib = t & 15u;
    const uint64_t x = w >> 8, invx = ~x & 0x00FFFFFFFFFFFFFFull;                     // 7 bytes after the token
    const uint32_t k = invx ? ((uint32_t)__ffsll((long long)invx) - 1) >> 3 : 7u;    // leading 255s
    const bool l15 = lnib == 15;
    easy = easy && (!l15 || k <= 5);
    const uint32_t ll = l15 ? 15 + 255 * k + ((uint32_t)(x >> (8 * (k & 7))) & 0xFFu) : lnib;
    const uint32_t q = p + 1 + (l15 ? k + 1 : 0);
    easy = easy && q + 15 < V.csize;                                                 // every length byte was readable (lz4.c:1986-2006)
    const uint32_t m = q + ll;
    easy = easy && m + 8 <= V.tlim && m + 16 <= V.csize;                             // not the last sequence; the match fields are resident
    const uint64_t y = easy ? tv_read8(V, m) : 0ull;
    const uint64_t z = y >> 16, invz = ~z & 0x0000FFFFFFFFFFFFull;                     // 6 bytes after the offset
    const uint32_t km = invz ? ((uint32_t)__ffsll((long long)invz) - 1) >> 3 : 6u;
    const bool m15 = mnib == 15;
    easy = easy && (!m15 || km <= 4);
    r.ll = ll; r.q = q; r.off = (uint32_t)y & 0xFFFFu;
    r.ml = (m15 ? 15 + 255 * km + ((uint32_t)(z >> (8 * (km & 7))) & 0xFFu) : mnib) + kMinMatch;
    r.nx = m + 2 + (m15 ? km + 1 : 0);
    if (!easy) r = tok_decode<true>(V, p);
    return r;
}

// Publish the records of lanes [0, nok) of a decoded batch (o / len / rec per lane), waiting for room in
// the record ring and for the copy to come within kOutAhead regions.  Turn holder only.
__device__ __forceinline__ void publish_batch(ParserS& S, char* smem, uint32_t nok, uint32_t o, uint32_t len, const SeqRec& rec) {
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    uint16_t* idx = (uint16_t*)(smem + kOffIdx);
    SeqRec* recs = (SeqRec*)(smem + kOffRecs);
    const uint32_t lane = lane_id();
    const uint32_t er = (o + len - 1) >> kRegionShift;
    uint32_t done = 0;
    bool fresh = false;                                   // the cached view of the copy's progress is good enough most of the time
    while (done < nok) {
        if (fresh) parser_refresh(S, smem);
        fresh = true;
        const unsigned long long okm = __ballot(lane >= done && lane < nok && er < S.g + kOutAhead);
        const unsigned long long run = ~(okm >> done);
        uint32_t npub = run ? (uint32_t)__ffsll((long long)run) - 1 : 64u;
        if (npub > nok - done) npub = nok - done;
        const uint32_t room = kRecCap - 1 - (S.head - S.tail);               // rows free, one kept for the sentinel
        if (npub > room) npub = room;
        if (npub == 0) { if (uload(&misc[M_ABORT])) return; spin_pause(); continue; }
        if (lane >= done && lane < done + npub) {
            const uint32_t j = S.head + (lane - done);
            recs[j & kRecMask] = rec;
            for (uint32_t g1 = (o + kRegion - 1) >> kRegionShift; (g1 << kRegionShift) < o + len; g1++) idx[g1 & kIdxMask] = (uint16_t)j;
            if (lane == done + npub - 1) recs[(j + 1) & kRecMask].outpos = o + len;     // sentinel: where the next record starts
        }
        const uint32_t oend = wave_readlane(o + len, done + npub - 1);
        wave_lds_fence();
        if (lane == 0) { lds_store_release(&misc[M_HEAD], S.head + npub); lds_store_release(&misc[M_EMIT], oend); }
        S.head += npub; S.obase = oend; done += npub;
    }
}
// One record from the slow path (every lane holds the same values).
__device__ __forceinline__ void publish_one(ParserS& S, char* smem, uint32_t litpos, uint32_t ll, uint32_t off, uint32_t len) {
    SeqRec rec; rec.outpos = S.obase; rec.litpos = litpos; rec.ll = ll; rec.off = off;
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    uint16_t* idx = (uint16_t*)(smem + kOffIdx);
    SeqRec* recs = (SeqRec*)(smem + kOffRecs);
    const uint32_t lane = lane_id(), o = S.obase;
    for (;;) {
        parser_refresh(S, smem);
        if (S.head + 2 - S.tail <= kRecCap && ((o + len) >> kRegionShift) < S.g + kOutAhead) break;
        if (uload(&misc[M_ABORT])) return;
        spin_pause();
    }
    if (lane == 0) { recs[S.head & kRecMask] = rec; recs[(S.head + 1) & kRecMask].outpos = o + len; }
    const uint32_t g1 = ((o + kRegion - 1) >> kRegionShift) + lane;            // len <= kSlowSpan: at most 9 regions
    if ((g1 << kRegionShift) < o + len) idx[g1 & kIdxMask] = (uint16_t)S.head;
    wave_lds_fence();
    if (lane == 0) { lds_store_release(&misc[M_HEAD], S.head + 1); lds_store_release(&misc[M_EMIT], o + len); }
    S.head += 1; S.obase = o + len;
}

// SLOW PATH (turn holder only): the sequence whose token is at S.e, whatever its size; every lane computes the
// same values, length fields are scanned 64 bytes at a time.  Same rules as the reference's safe loop.
// Returns 0: go on at S.e, 1: that was the block's last sequence, 2: malformed (reported) or the block is over.
__device__ __noinline__ int slow_token(ParserS& S, char* smem) {
    const uint8_t* cr = (const uint8_t*)(smem + kOffCr);
    const uint32_t lane = lane_id(), csize = S.csize, p = S.e;
    if (p >= csize) { parser_fail(smem, csize ? csize - 1 : 0); return 2; }
    S.ppos = p; set_clo(smem, p);
    if (!parser_wait_data(S, smem, p + 1)) return 2;
    const uint32_t t = cr[mod_cr(p)];
    uint32_t ll = t >> 4, q = p + 1;
    if (ll == 15) {
        for (;;) {
            if (!parser_wait_data(S, smem, q + 64)) return 2;
            const uint32_t pos = q + lane;
            const bool inb = pos + 15 < csize;                              // lz4.c:1986-2006: a length byte is read only there
            const uint32_t x = inb ? (uint32_t)cr[mod_cr(pos)] : 0u;
            const unsigned long long stopm = __ballot(!inb || x != 255);
            if (!stopm) {
                ll += 255 * 64; q += 64; S.ppos = q; set_clo(smem, q);
                if (ll > csize) { parser_fail(smem, p); return 2; }
                continue;
            }
            const uint32_t kk = (uint32_t)__ffsll((long long)stopm) - 1;
            if (!wave_readlane(inb ? 1u : 0u, kk)) { parser_fail(smem, p); return 2; }
            ll += 255 * kk + wave_readlane(x, kk); q += kk + 1;
            break;
        }
        if (ll > csize) { parser_fail(smem, p); return 2; }
    }
    const uint32_t rem = csize - q, room = S.capB - S.obase;
    const bool last = rem < ll + 8 || room < ll + kMfLimit;                 // lz4.c:2279
    if (last && (rem != ll || room < ll)) { parser_fail(smem, p); return 2; }   // lz4.c:2312-2318
    // the literal run, in records of at most kSlowSpan bytes (a last sequence always gets a record)
    if (ll || last) {
        uint32_t left = ll, lp = q;
        do {
            const uint32_t n = left < kSlowSpan ? left : kSlowSpan;
            publish_one(S, smem, lp, n, 0, n);
            left -= n; lp += n;
        } while (left);
    }
    if (last) return 1;
    const uint32_t m = q + ll;                                              // m + 8 <= csize
    S.ppos = m; set_clo(smem, m);                                           // (the copy reads literals from memory, not from the ring)
    if (!parser_wait_data(S, smem, m + 2)) return 2;
    const uint32_t off = (uint32_t)cr[mod_cr(m)] | ((uint32_t)cr[mod_cr(m + 1)] << 8);
    uint32_t ml = t & 15, nx = m + 2;
    if (ml == 15) {
        for (;;) {
            if (!parser_wait_data(S, smem, nx + 64)) return 2;
            const uint32_t pos = nx + lane;
            const uint32_t x = pos < csize ? (uint32_t)cr[mod_cr(pos)] : 0u;
            const bool badafter = pos + 5 > csize;                          // after a length byte at least 4 more bytes must follow
            const unsigned long long stopm = __ballot(x != 255 || badafter);
            if (!stopm) {
                ml += 255 * 64; nx += 64; S.ppos = nx; set_clo(smem, nx);
                if (ml > 0x7FFFFFF0u) { parser_fail(smem, p); return 2; }
                continue;
            }
            const uint32_t kk = (uint32_t)__ffsll((long long)stopm) - 1;
            ml += 255 * kk + wave_readlane(x, kk); nx += kk + 1;
            if (wave_readlane(badafter ? 1u : 0u, kk) || ml > 0x7FFFFFF0u) { parser_fail(smem, p); return 2; }
            break;
        }
    }
    ml += kMinMatch;
    const uint32_t ms = S.obase;
    if (off == 0 || off > ms - S.low) { parser_fail(smem, p); return 2; }   // lz4.c:2356
    if (S.capB - ms < ml + kLastLiterals) { parser_fail(smem, p); return 2; }   // lz4.c:2423
    {
        uint32_t left = ml;
        do {
            const uint32_t n = left < kSlowSpan ? left : kSlowSpan;
            publish_one(S, smem, nx, 0, off, n);
            left -= n;
        } while (left);
    }
    S.e = nx;
    return 0;
}

enum : uint32_t { OUT_NONE = 0, OUT_MERGE = 1, OUT_EXIT = 2, OUT_STOP = 3, OUT_OVER = 4 };

// The walkers' view of the chain: one stream byte per trip, no branches.  A lane is at a token (mode 0), inside a
// literal-length field (mode 1) or inside a match-length field (mode 2).  It stops (dead, at the token `tok`) exactly
// where tok_decode says "not inside the tile's reach": those tokens belong to the slow path.
struct WalkState { uint32_t p, tok, acc, mode, cnt; bool mlf, dead; };
__device__ __forceinline__ void walk_init(WalkState& s, uint32_t p) { s.p = p; s.tok = p; s.acc = 0; s.mode = 0; s.cnt = 0; s.mlf = false; s.dead = false; }
// consume byte b = stream[s.p] (the caller checked s.p < tlim)
__device__ __forceinline__ void walk_step(WalkState& s, uint32_t b, const TileView& V) {
    const bool m0 = s.mode == 0, m1 = s.mode == 1, m2 = s.mode == 2;
    const bool is255 = b == 255;
    const uint32_t ll0 = b >> 4;
    const bool ext0 = ll0 == 15;
    const bool litdone = (m0 && !ext0) || (m1 && !is255);                  // the literal length is complete with this byte
    const uint32_t ll = m0 ? ll0 : s.acc + b;
    const uint32_t m = s.p + 1 + ll;                                        // first byte after the literals
    const bool mlf = m0 ? (b & 15) == 15 : s.mlf;
    const bool cont = (m0 && ext0) || ((m1 || m2) && is255);               // the length field goes on
    const uint32_t cnt = m0 ? 0u : s.cnt + 1;
    const bool stop = (litdone && (m + 8 > V.csize || m + 2 > V.tlim)) || (cont && cnt >= kExtMax);
    s.tok = m0 ? s.p : s.tok;
    s.dead = stop;
    s.acc = m0 ? 15u : s.acc + b;
    s.mlf = mlf;
    s.cnt = litdone ? 0u : cnt;
    s.mode = litdone ? (mlf ? 2u : 0u) : (cont ? (m2 ? 2u : 1u) : 0u);
    s.p = litdone ? m + 2 : s.p + 1;
}

// what a lane knows after the speculative walk of its segment
struct LaneWalk { uint32_t x; uint32_t okind, opos, nb; };

__device__ __forceinline__ void sb_mark(uint32_t* sb, uint32_t rel) { atomicOr(&sb[rel >> 5], 1u << (rel & 31)); }
__device__ __forceinline__ bool sb_test(const uint32_t* sb, uint32_t rel) { return (sb[rel >> 5] >> (rel & 31)) & 1u; }

// P1 + P2.  Every lane walks the token chain of its segment from the segment's first byte - or, when the tile is
// (re)entered at a known true token `ent`, the lane of that segment from `ent` and the lanes below it not at all -
// marking the token positions it visits (P1); then walks on ("bridge") until it steps on a position a later lane
// marked, leaves the tile, or kBridgeTrips trips have passed (P2).  A wrong start walks over literal bytes misread
// as tokens (~6.5 B per step) and merges with the true chain after a few hundred bytes.
__device__ __forceinline__ LaneWalk walk_tile(const TileView& V, uint32_t* sb, uint32_t ent, uint32_t& trips) {
    const uint32_t lane = lane_id(), t0 = V.t0, t1 = t0 + kTile;
#pragma unroll
    for (uint32_t w = 0; w < kSeg / 32; w++) sb[lane * (kSeg / 32) + w] = 0;
    wave_lds_fence();
    const uint32_t seg_lo = t0 + lane * kSeg, seg_hi = seg_lo + kSeg;
    WalkState s; walk_init(s, seg_lo);
    bool idle = false;
    if (ent != kNone) { if (ent >= seg_hi) idle = true; else if (ent > seg_lo) walk_init(s, ent); }
    for (;;) {
        const bool run = !idle && !s.dead && !(s.mode == 0 && s.p >= seg_hi);
        if (!__any(run)) break;
        trips++;
        if (run) {
            if (s.p >= V.tlim) { s.tok = s.mode == 0 ? s.p : s.tok; s.dead = true; }      // ran off the block
            else {
                if (s.mode == 0) sb_mark(sb, s.p - t0);
                walk_step(s, tv_byte(V, s.p), V);
            }
        }
    }
    wave_lds_fence();
    LaneWalk L; L.x = s.p; L.okind = idle ? OUT_OVER : (s.dead ? OUT_STOP : OUT_NONE); L.opos = s.dead ? s.tok : s.p; L.nb = 0;
    for (uint32_t trip = 0;; trip++) {
        const bool run = L.okind == OUT_NONE;
        if (!__any(run)) break;
        trips++;
        if (run) {
            const bool at_tok = s.mode == 0;
            if (at_tok && s.p >= t1) { L.okind = OUT_EXIT; L.opos = s.p; }
            else if (at_tok && sb_test(sb, s.p - t0)) { L.okind = OUT_MERGE; L.opos = s.p; }
            else if (trip >= kBridgeTrips) { L.okind = OUT_OVER; L.opos = at_tok ? s.p : s.tok; }   // (a token either way)
            else if (s.p >= V.tlim) { L.okind = OUT_STOP; L.opos = at_tok ? s.p : s.tok; }
            else {
                L.nb += at_tok ? 1u : 0u;
                walk_step(s, tv_byte(V, s.p), V);
                if (s.dead) { L.okind = OUT_STOP; L.opos = s.tok; }
            }
        }
    }
    return L;
}

// P3 + P4: stitch the true chain through the walked tile and leave exactly its tokens in [cur, tend) in the bitmap.
//   spec == true : the tile was walked before its entry was known; the chain comes in at `cur` (any token of the tile):
//                  a serial walk from cur finds the first marked position, the lane that marked it is true from there on.
//   spec == false: the tile was walked from `cur` (walk_tile's ent): that lane is true from cur.
// A true lane's marks are true up to its exit; where its bridge merged into lane k's marks, lane k is true from there
// (<= 64 hops).  A bridge that did not merge ends the stitch at its last (true) token - never a wrong answer, only
// a shorter stretch.  stop: the token at tend needs the slow path.  stitch_chain is the serial part (the next tile's
// entry is known after it), stitch_marks the rest.
struct Stitch { uint32_t tend; bool stop; uint32_t myT, epos, ne; };
__device__ __forceinline__ Stitch stitch_chain(const TileView& V, const uint32_t* sb, const LaneWalk& L, uint32_t cur, bool spec, uint32_t& trips) {
    const uint32_t lane = lane_id(), t0 = V.t0, t1 = t0 + kTile;
    Stitch R; R.tend = cur; R.stop = true; R.myT = kNone; R.epos = kNone; R.ne = 0;
    uint32_t k = kNone, Tk = 0;
    bool chain = false;
    if (spec) {
        WalkState s; walk_init(s, cur);                      // every lane walks the same walk
        for (;;) {
            if (s.mode == 0) {
                if (s.p >= t1) { R.tend = s.p; R.stop = false; break; }
                if (sb_test(sb, s.p - t0)) { k = (s.p - t0) >> kSegShift; Tk = s.p; chain = true; break; }
                if (R.ne >= kEntrySteps) { R.tend = s.p; R.stop = false; break; }   // (a true token: the caller walks the rest of the tile from it)
                if (lane == R.ne) R.epos = s.p;
                R.ne++;
            }
            if (s.p >= V.tlim) { R.tend = s.mode == 0 ? s.p : s.tok; R.stop = true; break; }
            walk_step(s, tv_byte(V, s.p), V);
            trips++;
            if (s.dead) { R.tend = s.tok; R.stop = true; break; }
        }
    } else { k = (cur - t0) >> kSegShift; Tk = cur; chain = true; }
    if (chain) {
        for (uint32_t it = 0; it < 64; it++) {
            if (lane == k) R.myT = Tk;
            const uint32_t kind = wave_readlane(L.okind, k), pos = wave_readlane(L.opos, k);
            if (kind == OUT_MERGE) { k = (pos - t0) >> kSegShift; Tk = pos; continue; }
            R.tend = pos; R.stop = kind == OUT_STOP;
            break;
        }
    }
    return R;
}
__device__ __forceinline__ void stitch_marks(const TileView& V, uint32_t* sb, const LaneWalk& L, const Stitch& R, uint32_t& trips) {
    const uint32_t lane = lane_id(), t0 = V.t0, t1 = t0 + kTile, tend = R.tend;
    // ---- P4: drop the marks of the lanes the chain skipped and the marks before a lane's entry ...
    const bool active = R.myT != kNone;
    const uint32_t seg_lo = t0 + lane * kSeg;
#pragma unroll
    for (uint32_t w = 0; w < kSeg / 32; w++) {
        uint32_t v = sb[lane * (kSeg / 32) + w];
        const uint32_t base = seg_lo + 32 * w;
        if (!active) v = 0;
        else if (R.myT > base) v = (R.myT - base >= 32) ? 0u : (v & (0xFFFFFFFFu << (R.myT - base)));
        if (tend <= base) v = 0; else if (tend - base < 32) v &= (1u << (tend - base)) - 1u;
        sb[lane * (kSeg / 32) + w] = v;
    }
    wave_lds_fence();
    // ... add the bridges of the true lanes (walked again: they were not kept) ...
    {
        WalkState s; walk_init(s, L.x);
        uint32_t left = active ? L.nb : 0;
        while (__any(left != 0)) {
            trips++;
            if (left) {
                if (s.mode == 0) { if (s.p < tend && s.p < t1) sb_mark(sb, s.p - t0); left--; }
                if (left) walk_step(s, tv_byte(V, s.p), V);
            }
        }
    }
    // ... and the tokens of the entry walk
    if (lane < R.ne && R.epos < tend && R.epos < t1) sb_mark(sb, R.epos - t0);
    wave_lds_fence();
}

// Token list of a round: tokens [first, first + kTokRound) of the bitmap, in order.  Returns the tile's token count.
__device__ __forceinline__ uint32_t list_tokens(const uint32_t* sb, uint16_t* tl, uint32_t first) {
    const uint32_t lane = lane_id();
    uint32_t wv[kSeg / 32], cnt = 0;
#pragma unroll
    for (uint32_t w = 0; w < kSeg / 32; w++) { wv[w] = sb[lane * (kSeg / 32) + w]; cnt += (uint32_t)__popc(wv[w]); }
    const uint32_t incl = wave_incl_sum(cnt);
    const uint32_t N = wave_readlane(incl, 63);
    uint32_t k = incl - cnt;
#pragma unroll
    for (uint32_t w = 0; w < kSeg / 32; w++) {
        uint32_t v = wv[w];
        while (v) {
            const uint32_t b = (uint32_t)__ffs((int)v) - 1; v &= v - 1;
            if (k >= first && k - first < kTokRound) tl[k - first] = (uint16_t)(lane * kSeg + 32 * w + b);
            k++;
        }
    }
    wave_lds_fence();
    return N;
}

// hand-offs between tile owners
__device__ __forceinline__ void publish_entry(char* smem, uint32_t next_tile, uint32_t pos) {
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    if (lane_id() == 0) { lds_store_relaxed(&misc[M_ENT_POS], pos); lds_store_release(&misc[M_ENT_SEQ], next_tile); }
}
__device__ __forceinline__ bool wait_word(ParserS& S, char* smem, uint32_t word, uint32_t value) {
    const uint32_t* misc = (const uint32_t*)(smem + kOffMisc);
    if (uload(&misc[word]) == value) return true;
    const uint64_t t0 = clock_ticks();
    bool ok = true;
    for (;;) {
        if (uload(&misc[word]) == value) break;
        if (block_over(smem)) { ok = false; break; }
        spin_pause();
    }
    S.t_turn += clock_ticks() - t0;
    return ok;
}

// One tile.  Returns 0: done, go on with the owner's next tile; 1: the block is over (finished, failed, or somebody
// else's business).
__device__ __forceinline__ int parse_tile(ParserS& S, char* smem, uint32_t pw, uint32_t tile) {
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    uint32_t* sb = (uint32_t*)(smem + kOffPar + pw * kParBytes);
    uint16_t* tl = (uint16_t*)(smem + kOffPar + pw * kParBytes + kTile / 8);
    const uint32_t lane = lane_id(), csize = S.csize;
    const uint32_t t0 = tile * kTile, t1 = t0 + kTile;
    TileView V; V.cr = (const uint8_t*)(smem + kOffCr); V.t0 = t0; V.crT = mod_cr(t0); V.csize = csize;
    V.tlim = t1 + kLook < csize ? t1 + kLook : csize;
    if (!parser_wait_data(S, smem, V.tlim)) return 1;
    uint64_t tc = clock_ticks();
    bool have_turn = false, entry_sent = false, spec = true;
    uint32_t cur = kNone;                              // the true token the tile is (re)entered at; unknown during the first walk
    for (;;) {
        // ---- the walk: before the entry is known the first time, from a true token after a stop inside the tile
        const LaneWalk L = walk_tile(V, sb, cur, S.n_trips);
        { const uint64_t t = clock_ticks(); S.t_walk += t - tc; tc = t; }
        if (spec) {
            if (!wait_word(S, smem, M_ENT_SEQ, tile)) return 1;
            cur = uload(&misc[M_ENT_POS]);
            tc = clock_ticks();
            if (cur >= t1) break;                      // the chain jumps over this tile (a long literal run)
        }
        const Stitch St = stitch_chain(V, sb, L, cur, spec, S.n_trips);
        uint32_t tend = St.tend; bool stop = St.stop;
        // the next tile's owner can start stitching as soon as this tile's exit is known
        if (!entry_sent && !stop && tend >= t1) { publish_entry(smem, tile + 1, tend); entry_sent = true; }
        stitch_marks(V, sb, L, St, S.n_trips);
        spec = false;
        { const uint64_t t = clock_ticks(); S.t_stitch += t - tc; tc = t; }
        if (!have_turn) {
            if (!wait_word(S, smem, M_TURN, tile)) return 1;
            have_turn = true;
            S.obase = uload(&misc[M_EMIT]); S.head = uload(&misc[M_HEAD]);
            parser_refresh(S, smem);                   // (publish_batch trusts this view until it runs out of room)
            tc = clock_ticks();
        }
        // ---- P5: 64 tokens at a time - decode (the decoder, not the walk, is the authority: every sequence must start
        //      where its predecessor ended), place, apply the output-side rules (lz4.c:2279 as an error - a sequence
        //      inside a tile is never the last; 2356; 2423), publish
        uint32_t expect = cur;
        bool failed = false; uint32_t failpos = 0;
        for (uint32_t first = 0; !failed; first += kTokRound) {
            const uint32_t N = list_tokens(sb, tl, first);
            if (first >= N) break;
            const uint32_t nround = N - first < kTokRound ? N - first : kTokRound;
            bool cut = false;
            for (uint32_t base = 0; base < nround && !failed && !cut; base += 64) {
                const uint32_t i = base + lane;
                const bool have = i < nround;
                TokInfo ti; ti.ll = ti.q = ti.off = ti.ml = ti.nx = 0; ti.st = 0;
                uint32_t tp = 0;
                if (have) { tp = t0 + tl[i]; ti = tok_decode_fast(V, tp); }
                const uint32_t pnx = __shfl_up(ti.nx, 1u);
                const uint32_t want = lane ? pnx : expect;
                const uint32_t len = ti.ll + ti.ml;
                const uint32_t isum = wave_incl_sum(have ? len : 0u);
                const uint32_t o = S.obase + (isum - len);
                const bool chainbad = have && (ti.st != 0 || tp != want);
                const bool outbad = have && (o > S.capB || S.capB - o < ti.ll + kMfLimit || ti.off == 0 || ti.off > o + ti.ll - S.low
                                             || S.capB - (o + ti.ll) < ti.ml + kLastLiterals);
                const unsigned long long badm = __ballot(chainbad || outbad);
                uint32_t nok = nround - base < 64 ? nround - base : 64;
                if (badm) {
                    const uint32_t l = (uint32_t)__ffsll((long long)badm) - 1;
                    const uint32_t wl = wave_readlane(want, l), tpl = wave_readlane(tp, l), stl = wave_readlane(ti.st, l);
                    const bool cb = wave_readlane(chainbad ? 1u : 0u, l) != 0;
                    nok = l;
                    if (cb && !(tpl == wl && stl == 2)) {
                        // the list does not continue the chain here (or the token needs the slow path after all): the slow path takes the token the chain expects
                        cut = true; tend = wl; stop = true;
                        if (entry_sent) { failed = true; failpos = wl; }          // (cannot happen: the exit was already handed on)
                    } else { failed = true; failpos = tpl; }
                }
                SeqRec rec; rec.outpos = o; rec.litpos = ti.q; rec.ll = ti.ll; rec.off = ti.off;
                publish_batch(S, smem, nok, o, len, rec);
                if (nok) expect = wave_readlane(ti.nx, nok - 1);
            }
            if (cut) break;
        }
        { const uint64_t t = clock_ticks(); S.t_decode += t - tc; tc = t; }
        if (failed) { parser_fail(smem, failpos); return 1; }
        if (stop) {
            S.e = tend;
            const int rc = slow_token(S, smem);
            { const uint64_t t = clock_ticks(); S.t_slow += t - tc; tc = t; }
            if (rc == 1) { wave_lds_fence(); if (lane == 0) lds_store_release(&misc[M_FIN], 1u); return 1; }
            if (rc == 2) return 1;
            cur = S.e;
        } else cur = tend;
        if (cur >= t1) break;
    }
    // ---- the tile is done: entry of the next tile, turn, the stream below the next tile is free
    if (!entry_sent) publish_entry(smem, tile + 1, cur);
    if (!have_turn && !wait_word(S, smem, M_TURN, tile)) return 1;
    set_clo(smem, t1);
    wave_lds_fence();
    if (lane == 0) lds_store_release(&misc[M_TURN], tile + 1);
    return 0;
}

__device__ __forceinline__ void parser_role(uint32_t pw, uint32_t csize, uint32_t cap, uint32_t prefix, char* smem, uint64_t* prof) {
    wave_priority_high();
    ParserS S;
    S.csize = csize; S.capB = cap + kBias; S.low = kBias - prefix;
    S.e = 0; S.obase = kBias; S.head = 0; S.ppos = 0; S.g = kFirstRegion; S.tail = 0;
    S.t_wait = S.t_walk = S.t_stitch = S.t_decode = S.t_turn = S.t_slow = 0; S.n_trips = 0;
    for (uint32_t tile = pw; (uint64_t)tile * kTile < csize; tile += kParseWaves)
        if (parse_tile(S, smem, pw, tile)) break;
    if (prof && pw == 0 && lane_id() == 0) {
        prof[1] = S.t_wait | (S.t_turn << 32); prof[2] = S.t_walk; prof[3] = S.t_stitch; prof[4] = S.t_decode;
        prof[5] = S.t_slow | ((uint64_t)S.n_trips << 32);
    }
}

// ------------------------------------------------------------------------------ COPY
struct RegionCtx {
    char* smem;
    uint32_t R, x0, x1, slot;       // region, its output range, its ring slot
    uint32_t g;                     // regions below g are final
    lz4amd_gsrc src; uint32_t csize;   // the compressed block (literals are read from memory: the L2 still holds what the loader fetched)
    uint32_t ringB;                 // output position of ring address 0 two laps below the region
    uint32_t j0, nrec;              // records that overlap the region
    uint64_t mydone;                // chunks of this region that are final
};

__device__ __forceinline__ bool chunk_is_final(const RegionCtx& C, uint32_t c) {
    const uint32_t r = c >> 6;
    if (r < C.g) return true;
    if (r == C.R) return (C.mydone >> (c & 63)) & 1ull;
    int32_t s = (int32_t)C.slot - (int32_t)(C.R - r); if (s < 0) s += kSlots;
    const DoneEnt* e = (const DoneEnt*)(C.smem + kOffBits) + s;
    uint32_t tag; uint64_t mask;
    lds_load_tag_mask(&e->tag, &e->mask, tag, mask);         // the tag first: tag == want means the mask is this region's
    const uint32_t want = r + kSlots;
    return tag > want || (tag == want && ((mask >> (c & 63)) & 1ull));
}
// output bytes [sa, sb] final?  (sb - sa < 16)
__device__ __forceinline__ bool range_is_final(const RegionCtx& C, uint32_t sa, uint32_t sb) {
    if ((sb >> kRegionShift) < C.g) return true;
    bool ok = chunk_is_final(C, sa >> 4);
    if ((sb >> 4) != (sa >> 4)) ok = ok && chunk_is_final(C, sb >> 4);
    return ok;
}
__device__ __forceinline__ U32x4 ring_read16(const RegionCtx& C, uint32_t pos) {
    return lds_read16_at((const uint8_t*)(C.smem + kOffRing), ring_fold(pos - C.ringB));
}

// Bytes [lo, lo + n) of v := output bytes [d, d + n) of the piece (literal run or match of rec; ms = where
// the match starts).  false: a source is not final yet.  own_ok: every lower piece of d's own chunk is done
// (only a match with a period < 16 that starts inside a chunk reads its own chunk).
__device__ __forceinline__ bool item_fetch(const RegionCtx& C, bool is_lit, uint32_t d, uint32_t lo, uint32_t n,
                                           const SeqRec& rec, uint32_t ms, bool own_ok, U32x4& v) {
    if (is_lit) {
        const uint32_t A = rec.litpos + (d - rec.outpos);                  // stream position of output byte d; [A, A + n) is inside the block
        if (A >= lo && A - lo + 16 <= C.csize) v = ld_global16(C.src + (A - lo));
        else {                                                              // the block's first / last bytes: never read outside src[0, csize)
            v[0] = v[1] = v[2] = v[3] = 0;
#pragma nounroll
            for (uint32_t i = 0; i < n; i++) chunk_set_byte(v, lo + i, (uint32_t)C.src[A + i]);
        }
        return true;
    }
    uint32_t dist = rec.off;
    const uint32_t into = d - ms;
    if (into >= dist) {
        // the source lies inside this very match (it overlaps itself): every earlier period holds the same
        // bytes; read the farthest one the 64 KB window holds, long final, instead of the bytes just written
        uint32_t k = into / dist + 1;
        const uint32_t kmax = kMaxDistance / dist;
        if (k > kmax) k = kmax;
        dist *= k;
    }
    const uint32_t s = d - dist;
    // sources below my chunk must be final; sources inside my own chunk (a match that starts inside a chunk,
    // offset < 16 + lo) are in once every lower piece of the chunk is
    const uint32_t cstart = d - lo;
    const uint32_t se = n <= dist ? s + n - 1 : d - 1;                       // last source byte
    if (s < cstart && !range_is_final(C, s, se < cstart ? se : cstart - 1)) return false;
    if (se >= cstart && !own_ok) return false;
    if (n <= dist) {
        v = ring_read16(C, s - lo);
        return true;
    }
    // period < n <= 16: bytes [s, d) are the pattern
    const U32x4 pat = ring_read16(C, s);
    v[0] = v[1] = v[2] = v[3] = 0;
    uint32_t k = 0;
#pragma nounroll
    for (uint32_t i = 0; i < n; i++) { chunk_set_byte(v, lo + i, chunk_byte(pat, k)); if (++k == dist) k = 0; }
    return true;
}

__device__ __forceinline__ void lds_or16(char* smem, uint32_t off, const U32x4& v) {
    unsigned long long* q = (unsigned long long*)(smem + off);
    atomicOr(&q[0], (unsigned long long)v[0] | ((unsigned long long)v[1] << 32));
    atomicOr(&q[1], (unsigned long long)v[2] | ((unsigned long long)v[3] << 32));
}

// chunk c of the region's slot |= v (the ring's pad mirrors chunks 0-1 of slot 0 at all times, so that a 16-byte
// read that starts in the ring's last bytes runs on into valid data)
__device__ __forceinline__ void slot_or16(const RegionCtx& C, uint32_t c, const U32x4& v) {
    lds_or16(C.smem, kOffRing + (C.slot << kRegionShift) + (c << 4), v);
    if (C.slot == 0 && c < kRingPad / kChunk) lds_or16(C.smem, kOffRing + kRingBytes + (c << 4), v);
}
__device__ __forceinline__ void slot_write16(const RegionCtx& C, uint32_t c, const U32x4& v) {
    *(U32x4*)(C.smem + kOffRing + (C.slot << kRegionShift) + (c << 4)) = v;
    if (C.slot == 0 && c < kRingPad / kChunk) *(U32x4*)(C.smem + kOffRing + kRingBytes + (c << 4)) = v;
}

// round-B item of lane l in trip t: the head of a piece that starts inside a chunk
struct BItem { SeqRec rec; uint32_t ms, d, lo, n, chunk; bool is_lit, valid; };
__device__ __forceinline__ BItem b_item(const RegionCtx& C, uint32_t t) {
    const SeqRec* recs = (const SeqRec*)(C.smem + kOffRecs);
    const uint32_t l = lane_id(), r = 32 * t + (l >> 1);
    BItem it; it.valid = false; it.is_lit = !(l & 1);
    it.rec.outpos = it.rec.litpos = it.rec.ll = it.rec.off = 0; it.ms = it.d = it.lo = it.n = it.chunk = 0;
    if (r < C.nrec) {
        it.rec = recs[(C.j0 + r) & kRecMask];
        const uint32_t nout = recs[(C.j0 + r + 1) & kRecMask].outpos;
        it.ms = it.rec.outpos + it.rec.ll;
        it.d = it.is_lit ? it.rec.outpos : it.ms;
        const uint32_t pe = it.is_lit ? it.ms : nout;
        it.lo = it.d & 15u;
        it.valid = it.d >= C.x0 && it.d < C.x1 && it.lo != 0 && pe > it.d;
        const uint32_t n = pe - it.d;
        it.n = n < 16 - it.lo ? n : 16 - it.lo;
        it.chunk = (it.d - C.x0) >> 4;
    }
    return it;
}

// Compose region C.R in its ring slot and store it.
__device__ __forceinline__ void copy_region(RegionCtx& C, lz4amd_gdst dst, uint32_t w, uint64_t& t_retry) {
    char* smem = C.smem;
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    const SeqRec* recs = (const SeqRec*)(smem + kOffRecs);
    DoneEnt* ents = (DoneEnt*)(smem + kOffBits);
    unsigned long long* pend = (unsigned long long*)(smem + kOffPend) + w * (kMaxTrips + 2);     // [0] scratch, [1..] trips
    const uint32_t lane = lane_id();
    const uint32_t slot_off = kOffRing + (C.slot << kRegionShift);
    // the slot is mine now: no chunk of region R is done (mask first, then the tag)
    if (lane == 0) { lds_store_release64(&ents[C.slot].mask, 0ull); lds_store_release(&ents[C.slot].tag, C.R + kSlots); }
    C.mydone = 0;
    C.ringB = (C.R - C.slot - kSlots) << kRegionShift;
    // ---- round A: which record covers the first byte of each chunk?  (scratch: the slot itself)
    uint32_t* fs = (uint32_t*)(smem + slot_off);
    fs[lane] = 0;
    wave_lds_fence();
    for (uint32_t base = 1; base < C.nrec; base += 64) {
        const uint32_t r = base + lane;
        if (r < C.nrec) {
            const uint32_t o = recs[(C.j0 + r) & kRecMask].outpos;        // > x0
            const uint32_t s = (o - C.x0 + kChunk - 1) / kChunk;
            if (s < 64) atomicMax(&fs[s], r);
        }
    }
    wave_lds_fence();
    const uint32_t jr = wave_incl_max(fs[lane]);
    wave_lds_fence();
    const uint32_t c0 = C.x0 + kChunk * lane;
    const bool actA = c0 < C.x1;
    SeqRec arec; arec.outpos = arec.litpos = arec.ll = arec.off = 0;
    uint32_t ams = 0, an = 0; bool alit = false;
    if (actA) {
        arec = recs[(C.j0 + jr) & kRecMask];
        const uint32_t nout = recs[(C.j0 + jr + 1) & kRecMask].outpos;
        ams = arec.outpos + arec.ll;
        alit = c0 < ams;
        const uint32_t pe = alit ? ams : nout;
        an = pe - c0 < 16 ? pe - c0 : 16;
    }
    {
        U32x4 v; v[0] = v[1] = v[2] = v[3] = 0;
        bool ready = false;
        if (actA) {
            ready = item_fetch(C, alit, c0, 0, an, arec, ams, true, v);
            v = ready ? keep_bytes(v, 0, an) : U32x4{0, 0, 0, 0};
            slot_write16(C, lane, v);
        }
        uint64_t pendA = __ballot(actA && !ready);
        // ---- round B: heads of the pieces that start inside a chunk, 32 records per trip
        const uint32_t trips = (C.nrec + 31) / 32;
        bool anyB = false;
        for (uint32_t t = 0; t < trips; t++) {
            const BItem it = b_item(C, t);
            bool rdy = false;
            if (it.valid) {
                U32x4 bv;
                rdy = item_fetch(C, it.is_lit, it.d, it.lo, it.n, it.rec, it.ms, false, bv);
                if (rdy) slot_or16(C, it.chunk, keep_bytes(bv, it.lo, it.lo + it.n));
            }
            const unsigned long long pm = __ballot(it.valid && !rdy);
            if (lane == 0) pend[1 + t] = pm;
            anyB = anyB || pm != 0;
        }
        // ---- pieces whose sources were still in flight: try again until they are all in
        if (pendA || anyB) {
            const uint64_t tr0 = clock_ticks();
            uint64_t published = 0;
            const uint64_t actm = __ballot(actA);
            for (;;) {
                // chunks without a pending piece are final: tell the other waves
                wave_lds_fence();
                if (lane == 0) pend[0] = 0;
                wave_lds_fence();
                for (uint32_t t = 0; t < trips; t++) {
                    const unsigned long long pm = pend[1 + t];
                    if (pm) { const BItem it = b_item(C, t); if ((pm >> lane) & 1ull) atomicOr(&pend[0], 1ull << it.chunk); }
                }
                wave_lds_fence();
                const uint64_t pendchunks = pendA | pend[0];
                DTRACE("retry R=%u pendA=%llx pendB0=%llx chunks=%llx g=%u chi=%u\n", C.R, (unsigned long long)pendA, (unsigned long long)pend[1], (unsigned long long)pendchunks, C.g, C.chi);
                C.mydone = ~pendchunks;
                const uint64_t pub = ~pendchunks & actm;
                if (pub != published) { published = pub; if (lane == 0) lds_store_release64(&ents[C.slot].mask, pub); }
                if (!pendchunks) break;
                if (uload(&misc[M_ABORT])) return;
                spin_pause();
                C.g = first_open_region(smem);
                if (pendA) {
                    const bool mine = (pendA >> lane) & 1ull;
                    bool rdy = false;
                    if (mine) {
                        U32x4 av;
                        rdy = item_fetch(C, alit, c0, 0, an, arec, ams, true, av);
                        if (rdy) slot_or16(C, lane, keep_bytes(av, 0, an));
                    }
                    pendA &= ~__ballot(rdy);
                }
                bool earlier_clear = true;
                for (uint32_t t = 0; t < trips; t++) {
                    const unsigned long long pm = pend[1 + t];
                    if (!pm) continue;
                    const BItem it = b_item(C, t);
                    const bool mine = (pm >> lane) & 1ull;
                    // is every lower piece of my own chunk in? (chunk A, the earlier trips, the lower lanes of this trip)
                    const unsigned long long lower = pm & ((1ull << lane) - 1ull);
                    const uint32_t h = lower ? 63u - (uint32_t)__clzll((long long)lower) : 0u;
                    const uint32_t hc = (uint32_t)__shfl((int)it.chunk, (int)h);
                    const bool own_ok = earlier_clear && !((pendA >> it.chunk) & 1ull) && (!lower || hc != it.chunk);
                    bool rdy = false;
                    if (mine) {
                        U32x4 bv;
                        rdy = item_fetch(C, it.is_lit, it.d, it.lo, it.n, it.rec, it.ms, own_ok, bv);
                        if (rdy) slot_or16(C, it.chunk, keep_bytes(bv, it.lo, it.lo + it.n));
                    }
                    const unsigned long long left = pm & ~__ballot(rdy);
                    wave_lds_fence();
                    if (lane == 0) pend[1 + t] = left;
                    wave_lds_fence();
                    if (left) earlier_clear = false;
                }
            }
            t_retry += clock_ticks() - tr0;
        }
    }
    // ---- the region is complete: to HBM, then tell the other waves
    wave_lds_fence();
    if (actA) {
        const U32x4 v = *(const U32x4*)(smem + slot_off + kChunk * lane);
        const uint32_t c1 = c0 + kChunk < C.x1 ? c0 + kChunk : C.x1;
        if (c1 - c0 == kChunk) st_global16(dst + (c0 - kBias), v);
        else {
#pragma nounroll
            for (uint32_t i = 0; i < c1 - c0; i++) dst[c0 - kBias + i] = (uint8_t)chunk_byte(v, i);
        }
    }
    wave_lds_fence();
    if (lane == 0) lds_store_release64(&ents[C.slot].mask, ~0ull);
}

__device__ __forceinline__ void copy_role(uint32_t w, lz4amd_gsrc src, uint32_t csize, lz4amd_gdst dst, char* smem, uint64_t* prof) {
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    const uint16_t* idx = (const uint16_t*)(smem + kOffIdx);
    uint32_t* fin = (uint32_t*)(smem + kOffFin);
    const uint32_t lane = lane_id();
    uint32_t k = 0, R = kFirstRegion + w, slot = (kFirstRegion + w) % kSlots;
    uint64_t t_rec = 0, t_lead = 0, t_work = 0, t_retry = 0;
    for (;; R += kCopyWaves, slot = slot + kCopyWaves >= kSlots ? slot + kCopyWaves - kSlots : slot + kCopyWaves) {
        RegionCtx C; C.smem = smem; C.R = R; C.slot = slot; C.src = src; C.csize = csize;
        C.x0 = R << kRegionShift;
        // ---- wait until the records cover the region (or the block ends inside / before it)
        uint64_t ts = clock_ticks();
        uint32_t oe;
        for (;;) {
            oe = uload(&misc[M_EMIT]);
            if (oe >= C.x0 + kRegion) { C.x1 = C.x0 + kRegion; break; }
            if (uload(&misc[M_FIN])) {
                oe = uload(&misc[M_EMIT]);
                if (C.x0 >= oe) goto out;
                C.x1 = oe < C.x0 + kRegion ? oe : C.x0 + kRegion;
                break;
            }
            if (uload(&misc[M_ABORT])) goto out;
            spin_pause_long();
        }
        { const uint64_t t = clock_ticks(); t_rec += t - ts; ts = t; }
        // ---- flow control: region R takes the ring slot of region R-80, which regions up to R-16 may still read
        for (;;) {
            C.g = first_open_region(smem);
            if (C.g + kMaxLead >= R) break;
            if (uload(&misc[M_ABORT])) goto out;
            spin_pause();
        }
        { const uint64_t t = clock_ticks(); t_lead += t - ts; ts = t; }
        C.j0 = idx[R & kIdxMask];
        {
            const uint32_t head = uload(&misc[M_HEAD]);
            const uint32_t jl = (C.x1 < oe) ? (uint32_t)idx[(R + 1) & kIdxMask] : head - 1;
            uint32_t nrec = ((jl - C.j0) & 0xFFFFu) + 1;
            if (nrec > 32 * kMaxTrips) nrec = 32 * kMaxTrips;             // (more than 258 records never overlap a region)
            C.nrec = nrec;
        }
        uint64_t tr = 0;
        DTRACE("region R=%u x0=%u x1=%u j0=%u nrec=%u g=%u\n", R, C.x0, C.x1, C.j0, C.nrec, C.g);
        copy_region(C, dst, w, tr);
        DTRACE("region R=%u done\n", R);
        if (uload(&misc[M_ABORT])) goto out;
        k++;
        if (lane == 0) lds_store_release(&fin[w], k);
        { const uint64_t t = clock_ticks(); t_work += t - ts - tr; t_retry += tr; }
    }
out:
    if (prof && w == 0 && lane == 0) { prof[6] = t_rec | (t_lead << 32); prof[7] = t_work | (t_retry << 32); }
}

// ------------------------------------------------------------------------------ one block
__device__ __forceinline__ void decode_one_block(const DecBatch& P, uint32_t b, char* smem) {
    const uint32_t tid = threadIdx.x, w = wave_id();
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
    uint32_t prefix = P.prefix ? (uint32_t)P.prefix[b] : 0u; if (prefix > kBias) prefix = kBias;

    uint64_t* prof = P.prof ? P.prof + (uint64_t)blockIdx.x * 8 : nullptr;
    uint64_t tstart = 0;
    if (prof && tid == 0) tstart = clock_ticks();

    // -- control words, done entries, the history before dst (linked blocks, lz4.c:2719 usingDict prefix mode) -> ring
    if (tid == 0) {
        misc[M_ERR] = kNone; misc[M_ABORT] = 0; misc[M_FIN] = 0; misc[M_CHI] = 0; misc[M_CLO] = 0;
        misc[M_EMIT] = kBias; misc[M_HEAD] = 0; misc[M_ENT_SEQ] = 0; misc[M_ENT_POS] = 0; misc[M_TURN] = 0;
    }
    if (tid < 16) ((uint32_t*)(smem + kOffFin))[tid] = 0;
    if (tid < kSlots) { DoneEnt e; e.mask = 0; e.tag = 0; e.pad = 0; ((DoneEnt*)(smem + kOffBits))[tid] = e; }
    if (prefix) {
        uint8_t* ring = (uint8_t*)(smem + kOffRing);
        const uint32_t lo = kBias - prefix;
        for (uint32_t v = (lo & ~15u) + 16 * tid; v < kBias; v += 16 * kDecThreads) {
            U32x4 g; g[0] = g[1] = g[2] = g[3] = 0;
#pragma nounroll
            for (uint32_t i = 0; i < 16; i++) if (v + i >= lo) chunk_set_byte(g, i, (uint32_t)(dst - (kBias - (v + i)))[0]);
            *(U32x4*)(ring + v) = g;                         // positions below kBias sit at ring address = position
            if (v < kRingPad) *(U32x4*)(ring + kRingBytes + v) = g;
        }
    }
    __syncthreads();

    if (w == kLoadWave) loader_role(src, csize, smem);
    else if (w >= kCopyWaves) parser_role(w - kCopyWaves, csize, cap, prefix, smem, prof);
    else copy_role(w, src, csize, dst, smem, prof);

    __syncthreads();
    if (tid == 0) {
        P.result[b] = misc[M_ABORT] ? err_at(misc[M_ERR]) : (int32_t)(misc[M_EMIT] - kBias);
        if (prof) prof[0] = clock_ticks() - tstart;
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
