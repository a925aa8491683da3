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
// every probe.  Here ONE 1024-thread workgroup (16 waves, one CU, ~160 KB of its LDS) streams
// through a block in TILES of 8 KB, and everything the inner loops touch lives in LDS:
//
//   source ring   84 KB of the block around the tile (the 64 KB LZ4 window + the tile + the
//                 prefetched next tile), filled with coalesced 16-byte loads; candidates are
//                 verified and matches extended against it, never against HBM;
//   hash table    8192 x u32 (13 bits; 5 bytes hashed, 4 for blocks under 64 KB + 11 like lz4.c:1389),
//                 FROZEN while a tile is parsed: every position of the tile probes the state left
//                 by the previous tiles, and the tile's own positions are inserted afterwards with
//                 atomicMax (order independent, so the output is deterministic).  That decouples
//                 match FINDING from the parse order, which is what makes the tile parallel:
//   strips        the tile is cut in 16 strips of 512 bytes, one wave each (match_strip): a lane probes
//                 four positions of its own 8 source bytes; stretches of probe hits with one distance
//                 ("runs") are listed, measured lane = run (24 bytes on, 8 back over pending literals),
//                 and selected by wave scans - no serial walk over the matches; a long match is
//                 finished by a wave-wide compare.  A match may run past its strip up to the tile's end;
//   settle        one wave makes the strips of the tile BEFORE consistent (matches that ran over a
//                 strip's end: resolve_overruns) and turns their encoded sizes into output offsets
//                 (strip_offsets), while the others already parse the next tile;
//   emit          the settled tile's sequences are composed in a 12 KB staging buffer in LDS - token,
//                 length and offset bytes as byte writes, literals in 8-byte pieces numbered by a wave
//                 scan - by whichever wave is free (emit_strip_lds), and leave for HBM as whole aligned
//                 16-byte chunks while the next tile is inserted into the table (flush_begin / flush_end).
//                 The emit lanes also write the optional entry-point table's rows (hint_row; lz4amd_params.h).
//
// HBM traffic per block: source read once (plus the final literal run if it is longer than the
// ring), compressed stream written once, 16 bytes of table per ~512 bytes of source when a table is
// asked for.  No scratch, no second kernel.  No MFMA: byte shuffling.
#pragma once
#include "lz4_common.h"
#include "../lz4amd_params.h"

// what a wave does between two looks at a word it waits for inside a tile (the waits are a few thousand cycles long: a look costs the
// SIMD's other waves issue slots)
#ifndef LZ4AMD_CMP_WAIT_SLEEP
#define LZ4AMD_CMP_WAIT_SLEEP 3
#endif
#define CMP_WAIT_PAUSE() spin_pause_n<LZ4AMD_CMP_WAIT_SLEEP>()
#ifndef LZ4AMD_CMP_FLUSH_IN_A
#define LZ4AMD_CMP_FLUSH_IN_A 1    // developer knob: 0: the tile before leaves after the barrier, one chunk per thread
#endif
#ifndef LZ4AMD_CMP_ONE_BARRIER
#define LZ4AMD_CMP_ONE_BARRIER 1
#endif
#ifndef LZ4AMD_CMP_EARLY_COMMIT
#define LZ4AMD_CMP_EARLY_COMMIT 1
#endif
#ifndef LZ4AMD_CMP_RING
#define LZ4AMD_CMP_RING 82000
#endif
#ifndef LZ4AMD_CMP_IDLE_SETTLE
#define LZ4AMD_CMP_IDLE_SETTLE 1
#endif
#ifndef LZ4AMD_CMP_PRIO
#define LZ4AMD_CMP_PRIO 2          // developer knob: 1: the settling wave runs at high issue priority, 2: the measuring waves at raised priority
#endif
#ifndef LZ4AMD_CMP_EMIT_PRIO
#define LZ4AMD_CMP_EMIT_PRIO 1
#endif
#ifndef LZ4AMD_CMP_HASH32
#define LZ4AMD_CMP_HASH32 1
#endif
#ifndef LZ4AMD_CMP_MERGE_RUNS
#define LZ4AMD_CMP_MERGE_RUNS 1
#endif
#ifndef LZ4AMD_CMP_HASH_MUL
#define LZ4AMD_CMP_HASH_MUL 0x9E3779B9u      // (not the reference's 2654435761 = 0x9E3779B1, lz4.c:777: tests/datagen.c draws its bytes with a generator that multiplies by the very
                                             //  same constant, and the literals of `datagen -P0` then hash into a few table slots - same-address LDS atomics, 9.2 ms per 4 MiB block
                                             //  against 2.4 ms on uniform random bytes.  Measured on the GPU: 0x9E3779B9 and 0xCC9E2D51 2.4 ms, 0x27D4EB2F 3.8, 0x85EBCA6B / 0xC2B2AE35
                                             //  9.1, 0x165667B1 12.1; blocks made with 0xCC9E2D51 decode 9 % slower WITHOUT their tables (2.10 against 1.94 ms per GiB, by any build of
                                             //  the decoder; sequence statistics identical - not understood), with this one as with the reference's.  Compressible datagen: unchanged)
#endif
#ifndef LZ4AMD_STRIDE4_FROM
#define LZ4AMD_STRIDE4_FROM 2
#endif
#ifndef LZ4AMD_CMP_PROF
#if defined(LZ4AMD_PROF_ROLES) || defined(LZ4AMD_PROF_TILE) || defined(LZ4AMD_PROF_WAVES) || defined(LZ4AMD_PROF_HC)
#define LZ4AMD_CMP_PROF 1
#else
#define LZ4AMD_CMP_PROF 0        // 1: the phase stamps of tools/prof_cmp.py / prof_hc.py (LZ4AMD_PROF=1 in the environment): a developer build, tools/build_variant.sh prof -DLZ4AMD_CMP_PROF=1
#endif
#endif
namespace lz4amd {

// developer counters of the CPU interpreter's build (tools/exp/cmp_emu_stats.py): trips of the parse and emit loops
#ifdef LZ4AMD_EMU_STATS
extern "C" unsigned long long lz4amd_emu_stats[16];
#define CMP_STAT(i, v) do { if (lane_id() == 0) __atomic_fetch_add(&lz4amd_emu_stats[i], (unsigned long long)(v), __ATOMIC_RELAXED); } while (0)
#else
#define CMP_STAT(i, v) ((void)0)
#endif

using CompBatch = ::lz4amd_comp_params;   // argument block (lz4amd_params.h)

struct alignas(8) MatchRec { uint32_t ll; uint32_t mo; };   // mo = offset | (matchlen-4) << 16

enum : uint32_t {
    kCmpThreads = 1024,
    kCmpWaves = kCmpThreads / 64,
    kTileMax = 8192,                   // blocks >= 64 KB + 11 (which are probed at every second position)
    kTileMaxSmall = 2048,              // smaller blocks (4-byte hash, like the reference: lz4.c:1389)
    kTileMin = 1024,
    kStripMin = 256,
    kSrcRing = LZ4AMD_CMP_RING,                  // the 64 KB window + the tile that is parsed + the next tile (its granules go in while this one is parsed) + 16 bytes + 48 to spare (a match is extended 8 bytes back below the window)
    kSrcPad = 32,                      // mirror of the ring's first bytes: unaligned reads never wrap
    kHashBits = 13,
    kStrips = 16,                      // strips of a tile, one wave each
    kSettleWave = 0,                   // the wave that settles a tile (overrunning matches, strip sizes -> offsets)
    kRecsPerPair = 128,                // matches a pair's 1 KB strip may take (the rest of it becomes literals)
    kRecsPerStrip = kRecsPerPair / 2,  // ... a strip of a small tile
    kRecsPerTile = 8 * kRecsPerPair,   // record slots of a tile (8 pair strips or 16 small ones)
    kCandPerPass = 64,                 // match candidates (runs of probe positions with one distance) measured at a time
    kCandCap = 128,                    // ... listed per probe of a piece (512 bytes of datagen -P60 list ~32 of them, of -P90 ~100)
    kShortRun = 16,                    // literal runs up to this long are copied by the sequence's own lane
    kStageBytes = 12288,               // a tile's encoded bytes are composed here (LDS) and leave with 16-byte stores
    kMaxInput = 0x7E000000u,           // lz4.h:214 LZ4_MAX_INPUT_SIZE
    kSmallBlockLimit = 65536 + 11,     // lz4.c:710 LZ4_64Klimit
    kStripFields = 10,
};
// LDS carve-up (bytes)
enum : uint32_t {
    kCOffMisc = 0,                                        // u32[64]
    kCOffPair = kCOffMisc + 64 * 4,                       // u32[16]: a wave's list of the tile is complete (tile number << 16 | runs); u32[16]: [pair] its measuring wave reads the table no more (tile number)
    kCOffStrip = kCOffPair + 32 * 4,                      // u32[2][kStripFields][16] per-strip summaries (two tiles in flight)
    kCOffTab = kCOffStrip + 2 * kStripFields * kCmpWaves * 4,   // u32[1 << kHashBits]
    kCOffRecs = kCOffTab + (4u << kHashBits),             // MatchRec[2][kRecsPerTile]: a tile's strips behind one another (kRecsPerPair or kRecsPerStrip slots each)
    kCOffEnds = kCOffRecs + 2 * kRecsPerTile * 8,                   // u16[2][kRecsPerTile] where a record's match ends (from the strip's start)
    kCOffCandS = kCOffEnds + 2 * kRecsPerTile * 2,                  // u32[kCmpWaves][kCandCap] a candidate's first probe slot | distance << 8
    kCOffCandE = kCOffCandS + kCmpWaves * kCandCap * 4,             // u8[kCmpWaves][kCandCap] its last probe slot
    kCOffScr = kCOffCandE + kCmpWaves * kCandCap,                   // u32[kCmpWaves][64] the emit's scatter space
    kCOffCarry = kCOffScr + kCmpWaves * 64 * 4,                   // u8[2][16]: encoded bytes of the 16-byte chunk a tile's output ends in (they leave with the next tile)
    kCOffStage = kCOffCarry + 32,                                   // u8[kStageBytes]
    kCOffRing = kCOffStage + kStageBytes,
    kCmpLdsBytes = kCOffRing + kSrcRing + kSrcPad,
};
static_assert(kCmpLdsBytes <= 160u * 1024u, "LDS budget");
static_assert(kCOffRing % 16 == 0 && kCOffStage % 16 == 0 && kCOffCarry % 16 == 0 && kSrcRing % 16 == 0 && kCOffRecs % 8 == 0 && kCOffCandS % 4 == 0 && kCOffScr % 4 == 0, "LDS alignment");
static_assert(kSrcRing >= 65536 + (LZ4AMD_CMP_EARLY_COMMIT ? 2 : 1) * kTileMax + 16 + 48, "the ring holds the window of the tile that is parsed and the tile after it");
enum : uint32_t { CM_BLOCK = 0, CM_OUT = 1, CM_CARRY = 2, CM_FAIL = 3, CM_READY = 4,     // CM_READY: tiles whose output offsets are fixed
                  // work queues and counts of a tile, two sets by tile parity (a full tile has one barrier: a set is reset during the tile after):
                  CM_EMITQ = 48,       // + parity: next strip of the settled tile to write out (handed to whichever wave is free)
                  CM_INSQ = 50,        // next piece of a full tile to insert into the table (likewise)
                  CM_EMITDONE = 52,    // strips of the settled tile that are written out (a bit each)
                  CM_FLUSHQ = 54,      // next 64 chunks of the staging buffer to store (full tiles: whichever wave is free)
                  CM_SEQS = 6,         // sequences of the tiles settled so far
                  CM_HOVER = 7,        // entry-point table: a row did not fit the table's room (the table is then left invalid)
                  CM_ROWS = 32,        // ... rows of the tiles settled so far
                  CM_HTILE = 40,       // u32[2][4] per tile in flight: { first ordinal, first row, log2 of the row distance, - }
                  CM_INH = 56,         // u32[2] by tile parity: generation << 16 | distance of a taken match that ran into the tile's end (atomicMax): the next tile's first position tries it
                  CM_TILE = 8 };       // u32[2][4] per tile in flight: { first output position, end, written directly (not staged),
                                       //   first pending byte of its carry chunk }
enum : uint32_t { T_OUT0 = 0, T_OUT1 = 1, T_DIRECT = 2, T_CFROM = 3 };
enum : uint32_t { S_N = 0, S_ENC = 1, S_LL0 = 2, S_TAIL = 3, S_OUT = 4, S_CARRY = 5,
                  S_END = 6,      // where the strip's last match ends when it runs past the strip (else 0)
                  S_FIRST = 7,    // first record that is emitted (the ones before it were covered by an earlier strip's match)
                  S_P = 8,        // first source position the strip emits
                  S_ORD = 9 };    // sequences of the block before the strip's first one

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
// Not the reference's 32- / 64-bit multiplies (v_mul_lo_u32 runs at a quarter of the VALU rate on this chip) but
// full-rate 24-bit multiply-adds over the overlapping byte triples of the same bytes; any well mixed function of the
// bytes gives the same matches up to table collisions.
__device__ __forceinline__ uint32_t hash_pos32(uint32_t lo, uint32_t hi, bool small) {
#if LZ4AMD_CMP_HASH32
    // the form of the reference's 4-byte hash (lz4.c:777-783: one 32-bit multiply - measured on gfx950 it issues like a 24-bit one, tools/exp/valu_issue.hip),
    // the fifth byte added in with a 24-bit multiply-add: 4 instructions instead of 6
    // (blocks under 64 KB + 11 keep the 24-bit form over their four bytes: with it datagen -P90 in 64 KiB blocks comes out 1.5 % smaller)
    uint32_t h;
    if (small) h = __umul24(lo, 0x9E3779u) + __umul24(lo >> 8, 0x85EBCBu);
    else h = lo * LZ4AMD_CMP_HASH_MUL + __umul24(hi & 0xFFu, 0xC2B2AFu);
    return h >> (32 - kHashBits);
#else
    uint32_t h = __umul24(lo, 0x9E3779u) + __umul24(lo >> 8, 0x85EBCBu);
    if (!small) h += __umul24(hi & 0xFFu, 0xC2B2AFu);
    return h >> (32 - kHashBits);
#endif
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
__device__ __forceinline__ uint32_t ring_back(uint32_t o, uint32_t d) { const uint32_t x = o - d, y = x + kSrcRing; return x < y ? x : y; }      // (x wraps when o < d)
__device__ __forceinline__ uint32_t ring_fwd(uint32_t o, uint32_t d) { const uint32_t x = o + d, y = x - kSrcRing; return x < y ? x : y; }
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
    strip_len = t / kStrips; if (strip_len < kStripMin) strip_len = kStripMin;
}

// Where the strips of a tile lie: strip w covers [g0 + w * strip_len, g0 + (w + 1) * strip_len) cut to the tile [t0, t1); g0 is the
// tile's start.  (Measured and dropped: a grid that starts half a strip earlier in the largest tiles, so that the wave which also
// settles the tile before gets a half strip - a half strip costs nearly what a whole one does, 4.24 -> 4.46 ms per GiB.)
__device__ __forceinline__ uint32_t strip_origin(uint32_t t0, uint32_t tile_len, uint32_t strip_len) {
    (void)tile_len; (void)strip_len;
    return t0;
}
__device__ __forceinline__ uint32_t strip_lo(uint32_t g0, uint32_t t0, uint32_t w, uint32_t strip_len) {
    const uint32_t c = g0 + w * strip_len; return c > t0 ? c : t0;
}

// ------------------------------------------------------------------------------ match (one strip)
// 8 bytes at ring offset o (any alignment) as two dwords: three ALIGNED dword reads and two v_alignbyte (a misaligned LDS
// access of any width is serialised lane by lane on gfx950: 64 cycles instead of ~3, tools/exp/lds_prims.hip)
struct Pair32 { uint32_t lo, hi; };
__device__ __forceinline__ Pair32 ring_ld8_32(const uint8_t* ring, uint32_t o) {
    const uint32_t* a = (const uint32_t*)(ring + (o & ~3u));
    const uint32_t sh = o & 3u;
    const uint32_t d0 = a[0], d1 = a[1], d2 = a[2];
    Pair32 r; r.lo = align_bytes(d1, d0, sh); r.hi = align_bytes(d2, d1, sh);
    return r;
}

// Matching a strip.  SH = 1: every second position is probed (blocks >= 64 KB + 11), SH = 0: every position, SH = 2: every fourth.
//   probe     a lane owns 4 << SH consecutive source bytes = four probe positions: its bytes come out of ONE aligned
//             read, the four hashes out of registers, each candidate is checked for its first four bytes only;
//   runs      consecutive probe positions that hit with the SAME distance are one match seen several times: only the
//             first and the last probe of such a run matter (where the match may start, up to where it is known to
//             hold).  Run starts / ends are numbered by a wave scan and written to a list (probe_list);
//   measure   lane = run: how far does it go on behind its last probe (24 bytes on its own; the rare longer ones are
//             finished by the whole wave when the run is taken), how far back before its first probe (lz4.c:1105-1109);
//   select    greedy in position order (a run that starts inside the match taken before it is cut to start at that
//             match's end - the reference would probe there and find the same distance), by wave scans;
//   records   the taken lanes write {literals, offset, length} and their encoded sizes side by side (parse_pass).
// Who does what: in the small tiles at a block's start and in small blocks one wave does all of it for its strip
// (match_strip).  In a full 8 KB tile of a big block the sixteen waves probe and list 512 bytes each, and the two waves of
// a PAIR then split up: one measures / selects / records the pair's 1 KB as ONE strip - the two lists behind one another,
// 64 runs at a time: at ~32 runs per 512 bytes that is one full pass where two waves ran a half-empty one each - while
// the other writes out strips of the tile before (match_pair_strip; the roles swap from tile to tile).
// one probe round: the four positions of a lane's 4 << SH source bytes from q0 on
struct Round { uint32_t dd[4]; uint32_t sb, eb; uint32_t hh[2]; };      // distance of the candidate that holds at slot j (0: none); run starts / ends; the four table indices probed (two per word)
template <bool SMALL>
__device__ __forceinline__ Round probe_round(const uint8_t* ring, const uint32_t* tab, uint32_t o0, uint32_t q0, uint32_t q_hi, uint32_t SH, uint32_t inh = 0u) {      // inh: a distance to try at q0 when the table has none (a tile's first position: the match that ran into the end of the tile before)
    constexpr bool small = SMALL;      // (SH: 0 for a small block; 1 or 2 - wave-uniform, the same for the whole launch - for a big one: only the bytes' extraction differs)
    Round R;
    uint32_t R0, R1, R2 = 0, R3 = 0, R4 = 0;
    if (!SMALL && SH == 2) { const U32x4 v = *(const U32x4*)(ring + o0); R0 = v[0]; R1 = v[1]; R2 = v[2]; R3 = v[3]; R4 = *(const uint32_t*)(ring + o0 + 16); }      // (o0 is a multiple of 16)
    else if (!SMALL) { const uint64_t v = *(const uint64_t*)(ring + o0); R0 = (uint32_t)v; R1 = (uint32_t)(v >> 32); R2 = *(const uint32_t*)(ring + o0 + 8); }
    else { R0 = *(const uint32_t*)(ring + o0); R1 = *(const uint32_t*)(ring + o0 + 4); }
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t q = q0 + (j << SH);
        uint32_t f0, f4 = 0;
        if (!SMALL && SH == 2) {                        // every fourth position: whole dwords
            f0 = j == 0 ? R0 : j == 1 ? R1 : j == 2 ? R2 : R3;
            f4 = j == 0 ? R1 : j == 1 ? R2 : j == 2 ? R3 : R4;
        } else if (!SMALL) {
            f0 = j == 0 ? R0 : j == 1 ? align_bytes(R1, R0, 2) : j == 2 ? R1 : align_bytes(R2, R1, 2);
            f4 = j == 0 ? R1 : j == 1 ? R1 >> 16 : j == 2 ? R2 : R2 >> 16;
        } else f0 = j == 0 ? R0 : align_bytes(R1, R0, j);
        const uint32_t hj = hash_pos32(f0, f4, small);
        if (j & 1) R.hh[j >> 1] |= hj << 16; else R.hh[j >> 1] = hj;
        const uint32_t c = tab[hj];
        const uint32_t d = q - c;
        const bool ok = q <= q_hi && d <= kMaxDistance;                 // q - c <= 65535 (c < q: the table is frozen while a tile is probed - it holds positions of earlier tiles, or 0)
        const uint32_t co = ok ? ring_back(o0 + (j << SH), d) : 0u;
        const uint32_t* a = (const uint32_t*)(ring + (co & ~3u));
        const uint32_t x = align_bytes(a[1], a[0], co & 3u);
        R.dd[j] = (ok && x == f0) ? d : 0u;
        if (j == 0 && inh && R.dd[0] == 0u && q <= q_hi) {
            // a long match is found ONCE by a serial parse (lz4.c:1104: it runs on to the block's end); a tile has to find it again, and the table's slot
            // of a position 64 K back has been taken many times over since: the tile's first position also tries the distance the tile before ended with
            const uint32_t ci = ring_back(o0, inh);
            const uint32_t* ai = (const uint32_t*)(ring + (ci & ~3u));
            if (align_bytes(ai[1], ai[0], ci & 3u) == f0) R.dd[0] = inh;
        }
    }
    // runs: first / last probe of every stretch of hits with one distance
#if LZ4AMD_CMP_MERGE_RUNS
    if (SMALL || SH <= 1) {
        // ... and a run goes on over ONE probe that missed (the table's slot was taken by something else) when the probe behind it hits with
        // the run's distance again: the four bytes either probe compared lie next to each other (every second position) or overlap
        // (every position) - the bytes in between are known to match.  Inside a long match most runs are such fragments.
        const uint32_t D0 = R.dd[0], D1 = R.dd[1], D2 = R.dd[2], D3 = R.dd[3];
        const uint32_t m1 = wave_prev_u32(D3), m2 = wave_prev_u32(D2), p4 = wave_next_u32(D0), p5 = wave_next_u32(D1);
#define LZ4AMD_GOES_ON(x, y, z) ((x) == (y) || ((y) == 0u && (x) == (z)))      // slot x's run continues into the neighbour y, or over the missed neighbour into z
        R.sb = ((D0 && !LZ4AMD_GOES_ON(D0, m1, m2)) ? 1u : 0u) | ((D1 && !LZ4AMD_GOES_ON(D1, D0, m1)) ? 2u : 0u) |
               ((D2 && !LZ4AMD_GOES_ON(D2, D1, D0)) ? 4u : 0u) | ((D3 && !LZ4AMD_GOES_ON(D3, D2, D1)) ? 8u : 0u);
        R.eb = ((D0 && !LZ4AMD_GOES_ON(D0, D1, D2)) ? 1u : 0u) | ((D1 && !LZ4AMD_GOES_ON(D1, D2, D3)) ? 2u : 0u) |
               ((D2 && !LZ4AMD_GOES_ON(D2, D3, p4)) ? 4u : 0u) | ((D3 && !LZ4AMD_GOES_ON(D3, p4, p5)) ? 8u : 0u);
#undef LZ4AMD_GOES_ON
        return R;
    }
#endif
    const uint32_t pv = wave_prev_u32(R.dd[3]), nx = wave_next_u32(R.dd[0]);
    R.sb = ((R.dd[0] && R.dd[0] != pv) ? 1u : 0u) | ((R.dd[1] && R.dd[1] != R.dd[0]) ? 2u : 0u) |
           ((R.dd[2] && R.dd[2] != R.dd[1]) ? 4u : 0u) | ((R.dd[3] && R.dd[3] != R.dd[2]) ? 8u : 0u);
    R.eb = ((R.dd[0] && R.dd[0] != R.dd[1]) ? 1u : 0u) | ((R.dd[1] && R.dd[1] != R.dd[2]) ? 2u : 0u) |
           ((R.dd[2] && R.dd[2] != R.dd[3]) ? 4u : 0u) | ((R.dd[3] && R.dd[3] != nx) ? 8u : 0u);
    return R;
}
// the runs of one round whose numbers fall into [0, kCandCap) (iS / iE: number of the lane's first start / end, already
// relative to the pass) into the list: candS[i] = the run's first slot (4 * lane + j: its position is the slot << SH bytes from
// the piece's start) | distance << 8, candE[i] = its last slot.
// slot number / distance of the lowest run start or end in a 4-bit mask, by bit tests (an index computed with ffs makes
// the compiler keep dd[] in scratch memory and load from it: a trip to HBM in the middle of the match)
__device__ __forceinline__ uint32_t slot_of(uint32_t low) { return (low >> 1) - (low >> 3); }           // 1, 2, 4, 8 -> 0, 1, 2, 3
__device__ __forceinline__ uint32_t dist_of(const Round& R, uint32_t low) { return (low & 1u) ? R.dd[0] : (low & 2u) ? R.dd[1] : (low & 4u) ? R.dd[2] : R.dd[3]; }
__device__ __forceinline__ void list_round(const Round& R, uint32_t iS, uint32_t iE, uint32_t slot0, uint32_t* candS, uint8_t* candE) {
    uint32_t s = R.sb, e2 = R.eb;
    {   // a lane's first start and first end (nearly always its only ones) without a trip around the loop
        const uint32_t ls = s & (0u - s), le = e2 & (0u - e2);
        if (s && iS < kCandCap) candS[iS] = (slot0 + slot_of(ls)) | (dist_of(R, ls) << 8);
        if (e2 && iE < kCandCap) candE[iE] = (uint8_t)(slot0 + slot_of(le));
        iS += s ? 1u : 0u; iE += e2 ? 1u : 0u;
        s ^= ls; e2 ^= le;
    }
    while (__any((s | e2) != 0)) {
        CMP_STAT(5, 1);
        const uint32_t ls = s & (0u - s), le = e2 & (0u - e2);
        if (s && iS < kCandCap) candS[iS] = (slot0 + slot_of(ls)) | (dist_of(R, ls) << 8);
        if (e2 && iE < kCandCap) candE[iE] = (uint8_t)(slot0 + slot_of(le));
        iS += s ? 1u : 0u; iE += e2 ? 1u : 0u;
        s ^= ls; e2 ^= le;
    }
}
// probe the piece [cs, cs + (256 << SH)) (positions up to q_hi) and list its runs [lo, lo + kCandCap) in position order; returns the
// number of runs the piece has.  probe_h: the table indices of the lane's four positions (the insert of this tile uses them again).
template <bool SMALL>
__device__ __forceinline__ uint32_t probe_list(const uint8_t* ring, const uint32_t* tab, uint32_t* candS, uint8_t* candE, uint32_t cs, uint32_t cs_off,
                                               uint32_t q_hi, uint32_t lo, uint32_t SH, uint32_t (&probe_h)[2], uint32_t inh_pos = 0xFFFFFFFFu, uint32_t inh_d = 0u) {
    const uint32_t lane = lane_id();
    CMP_STAT(6, 1);
    const Round A = probe_round<SMALL>(ring, tab, ring_fwd(cs_off, lane * (4u << SH)), cs + lane * (4u << SH), q_hi, SH, cs + lane * (4u << SH) == inh_pos ? inh_d : 0u);
    probe_h[0] = A.hh[0]; probe_h[1] = A.hh[1];
    const uint32_t cntA = (uint32_t)__popc(A.sb) | ((uint32_t)__popc(A.eb) << 16);
    const uint32_t inclA = wave_incl_sum(cntA), exA = inclA - cntA;
    list_round(A, (exA & 0xFFFFu) - lo, (exA >> 16) - lo, 4 * lane, candS, candE);      // (indices below lo wrap and are dropped)
    return wave_readlane(inclA, 63) & 0xFFFFu;
}

// what a strip's parse carries from pass to pass
struct ParseState { uint32_t nseq, enc, ll0, cur; };      // records so far, their encoded bytes, the first one's literals, end of the last match taken (first byte not yet covered)
// One pass of measure / select / records, lane = run: the run's first probe position qs, its distance d, a = the first byte
// its probes did not compare (last probe + 4).  Runs come in position order; st.cur carries the end of what was taken before.
__device__ __forceinline__ void parse_pass(const uint8_t* ring, MatchRec* recs, uint16_t* ends, uint32_t rec_cap, uint32_t cs, uint32_t cs_off,
                                           uint32_t mlimit, uint32_t last_q, bool have, uint32_t qs, uint32_t d, uint32_t a, ParseState& st, uint32_t* inh_w = nullptr, uint32_t inh_gen = 0u) {
    const uint32_t lane = lane_id();
    CMP_STAT(0, 1); { const unsigned long long hv = __ballot(have); CMP_STAT(1, __popcll(hv)); }
    // ---- measure
    uint32_t e = 0, back = 0, more = 0;
    if (have) {
        e = mlimit;
        if (a < mlimit) {
            // the next 24 bytes on the lane's own (seven aligned dwords per side)
            const uint32_t ao = ring_fwd(cs_off, a - cs), ko = ring_back(ao, d);
            const uint32_t* xa = (const uint32_t*)(ring + (ao & ~3u));
            const uint32_t* ya = (const uint32_t*)(ring + (ko & ~3u));
            uint32_t xd[7], yd[7];
#pragma unroll
            for (uint32_t i = 0; i < 7; i++) { xd[i] = xa[i]; yd[i] = ya[i]; }
            // (the first dword that differs is picked by a chain of selects, its first differing byte found once)
            uint32_t zf = 0, base = 24;
#pragma unroll
            for (int i = 5; i >= 0; i--) {
                const uint32_t z = align_bytes(xd[i + 1], xd[i], ao & 3u) ^ align_bytes(yd[i + 1], yd[i], ko & 3u);
                zf = z ? z : zf; base = z ? 4 * (uint32_t)i : base;
            }
            uint32_t same = zf ? base + ((uint32_t)(__ffs((int)zf) - 1) >> 3) : 24u;
            more = (same == 24 && a + 24 < mlimit) ? 1u : 0u;
            if (same > mlimit - a) same = mlimit - a;
            e = a + same;
        }
        // how far back, over literals that may still be pending (lz4.c:1105-1109)?
        if (qs - d >= 8) {
            const uint32_t bo = ring_back(ring_fwd(cs_off, qs - cs), 8);
            const Pair32 v = ring_ld8_32(ring, bo), k = ring_ld8_32(ring, ring_back(bo, d));
            const uint32_t x1 = v.hi ^ k.hi, x0 = v.lo ^ k.lo;
            back = x1 ? (uint32_t)__clz((int)x1) >> 3 : (x0 ? 4 + ((uint32_t)__clz((int)x0) >> 3) : 8u);
        }
    }
    // ---- select: a run is taken when it still holds a minimal match behind everything that ends before it - the
    //      running maximum of the ends of the runs in front of it (a wave scan: no serial walk over the matches; a
    //      run that is not taken ends less than 4 bytes behind a taken one, so the maximum over all runs is the end
    //      of the last taken match give or take 3 bytes).  A taken run that went on matching for more than the 24
    //      bytes its lane compared is finished by the whole wave, and the scan is repeated with its real end
    //      (finishing every such run before the scan was measured: slower, most of them end up covered).
    const bool sv = have && e >= qs + kMinMatch;
    unsigned long long taken;
    for (;;) {
        uint32_t pmx = wave_prev_u32(wave_incl_max(sv ? e : 0u));
        if (pmx < st.cur) pmx = st.cur;
        const uint32_t s0 = qs > pmx ? qs : pmx;
        bool tk = sv && e >= s0 + kMinMatch && s0 <= last_q;
        taken = __ballot(tk);
        const uint32_t room = rec_cap - st.nseq;
        if ((uint32_t)__popcll(taken) > room) taken = __ballot(tk && lanes_below(taken) < room);
        const unsigned long long mm = taken & __ballot(more != 0);
        if (!mm) break;
        CMP_STAT(2, 1);
        // long match: wave-wide compare, 8 bytes per lane per trip (lz4.c:680-703 LZ4_count)
        const uint32_t l = (uint32_t)__ffsll((long long)mm) - 1;
        uint32_t el = wave_readlane(e, l);
        const uint32_t eo = ring_fwd(cs_off, el - cs), ko = ring_back(eo, wave_readlane(d, l));
        uint32_t ext = 0;
        for (;;) {
            const uint32_t aa = el + ext + 8 * lane;
            uint32_t same = 0;
            if (aa < mlimit) {
                const Pair32 x = ring_ld8_32(ring, ring_fwd(eo, ext + 8 * lane)), y = ring_ld8_32(ring, ring_fwd(ko, ext + 8 * lane));
                same = equal_bytes8_32(x.lo, x.hi, y.lo, y.hi);
                if (same > mlimit - aa) same = mlimit - aa;
            }
            const unsigned long long brk = __ballot(same < 8);
            if (brk) {
                const uint32_t fl = (uint32_t)__ffsll((long long)brk) - 1;
                ext += 8 * fl + wave_readlane(same, fl);
                break;
            }
            ext += 512; CMP_STAT(3, 1);
        }
        el += ext;
        e = lane == l ? el : e;
        more = lane == l ? 0u : more;
    }
    if (!taken) return;
    const uint32_t ntaken = (uint32_t)__popcll(taken);
    CMP_STAT(4, ntaken);
    const bool mine = (taken >> lane) & 1;
    const uint32_t pt = wave_incl_max(mine ? e : 0u);
    uint32_t prev_end = wave_prev_u32(pt);                   // end of the match taken before mine (taken lanes)
    if (prev_end < st.cur) prev_end = st.cur;
    const uint32_t last_e = wave_readlane(pt, 63);
    if (inh_w && mine && e >= mlimit) atomicMax(inh_w, (inh_gen << 16) | d);      // (CM_INH: order independent - the output stays a function of the input)
    // ---- records
    const uint32_t ri = st.nseq + lanes_below(taken);
    uint32_t my_enc = 0, my_ll = 0;
    uint32_t my_mo = 0;
    if (mine) {
        uint32_t start = prev_end;
        if (qs > prev_end) { uint32_t bk = back; if (bk > qs - prev_end) bk = qs - prev_end; start = qs - bk; }
        my_ll = start - prev_end;
        const uint32_t mlen = e - start;
        my_mo = d | ((mlen - kMinMatch) << 16);
        my_enc = enc_size(my_ll, mlen - kMinMatch);
    }
    if (st.nseq == 0) st.ll0 = wave_readlane(my_ll, (uint32_t)__ffsll((long long)taken) - 1);
    const uint32_t enc_incl = wave_incl_sum(my_enc);
    if (mine) {
        // (ll's upper half: the encoded bytes of the strip's records before this one - what the settle subtracts when an earlier strip's
        //  match covers the strip's first records; a tile's literal runs and a strip's encoded bytes are far below 64 K)
        MatchRec r; r.ll = my_ll | ((st.enc + enc_incl - my_enc) << 16); r.mo = my_mo;
        recs[ri] = r;
        ends[ri] = (uint16_t)(e - cs);
    }
    st.enc += wave_readlane(enc_incl, 63);
    st.nseq += ntaken;
    if (last_e > st.cur) st.cur = last_e;
}

// the strip's summary for the settle (lane 0)
__device__ __forceinline__ void strip_summary(uint32_t* strip, uint32_t si, const ParseState& st, uint32_t ce) {
    if (lane_id() == 0) {
        strip[S_N * kCmpWaves + si] = st.nseq;
        strip[S_ENC * kCmpWaves + si] = st.enc;
        strip[S_LL0 * kCmpWaves + si] = st.ll0;
        strip[S_TAIL * kCmpWaves + si] = st.cur < ce ? ce - st.cur : 0;
        strip[S_END * kCmpWaves + si] = st.cur > ce ? st.cur : 0;
    }
}
// One wave parses the piece [pcs, pce) of the strip that starts at cs, all by itself: probe, list, and passes over the list; more
// runs than the list holds are rare: the piece is simply probed again for the next ones (the table is frozen: same answers).
template <bool SMALL>
__device__ __forceinline__ void parse_piece(const uint8_t* ring, const uint32_t* tab, MatchRec* recs, uint16_t* ends, uint32_t rec_cap,
                                            uint32_t* candS, uint8_t* candE, uint32_t cs, uint32_t pcs, uint32_t pce, uint32_t mlimit, uint32_t last_q,
                                            uint32_t (&probe_h)[2], uint32_t SH, ParseState& st) {
    const uint32_t lane = lane_id();
    const uint32_t cs_off = src_ring_off(cs), pcs_off = ring_fwd(cs_off, pcs - cs);
    const uint32_t q_hi = pce - 1 < last_q ? pce - 1 : last_q;   // last position of the piece that may start a match
    for (uint32_t lo = 0; st.nseq < rec_cap; lo += kCandCap) {
        uint32_t ph[2];
        const uint32_t total = probe_list<SMALL>(ring, tab, candS, candE, pcs, pcs_off, q_hi, lo, SH, ph);
        if (lo == 0) { probe_h[0] = ph[0]; probe_h[1] = ph[1]; }
        if (lo >= total) break;
        wave_lds_fence_local();
        const uint32_t nlist = total - lo < kCandCap ? total - lo : kCandCap;
        for (uint32_t pl = 0; pl < nlist && st.nseq < rec_cap; pl += kCandPerPass) {
            const bool have = pl + lane < nlist;
            uint32_t S = 0, E = 0;
            if (have) { S = candS[pl + lane]; E = candE[pl + lane]; }
            parse_pass(ring, recs, ends, rec_cap, cs, cs_off, mlimit, last_q, have, pcs + ((S & 255u) << SH), have ? S >> 8 : 1u, pcs + (E << SH) + kMinMatch, st);
        }
        if (lo + kCandCap >= total) break;
        wave_lds_fence_local();                                            // (the list is rewritten by the next probe)
    }
}
// ... a whole strip [cs, ce) of at most 256 << SH bytes: strip `w` of its tile (records, summary)
template <bool SMALL>
__device__ __forceinline__ void match_strip(const uint8_t* ring, const uint32_t* tab, MatchRec* recs, uint16_t* ends, uint32_t rec_cap, uint32_t* strip,
                                            uint32_t* candS, uint8_t* candE, uint32_t w, uint32_t n, uint32_t cs, uint32_t ce, uint32_t tend, uint32_t (&probe_h)[2], uint32_t SH) {
    ParseState st; st.nseq = 0; st.enc = 0; st.ll0 = 0; st.cur = cs;
    // positions that may start a match: q <= n - 12; matches end <= n - 5 and <= tend
    if (n >= kMfLimit + 1 && cs <= n - kMfLimit) {
        // a match may run past the strip up to the tile's end: the strips it covers give way (resolve_overruns)
        uint32_t mlimit = n - kLastLiterals; if (mlimit > tend) mlimit = tend;
        parse_piece<SMALL>(ring, tab, recs, ends, rec_cap, candS, candE, cs, cs, ce, mlimit, n - kMfLimit, probe_h, SH, st);
    }
    strip_summary(strip, w, st, ce);
}
// The measuring wave of a pair: the pair's 1 KB [cs, cs + 1024) as one strip (number `si` of its tile).  Its two pieces were probed and
// listed by the pair's two waves (nA / nB runs, lists LA / LB: the first piece's wave is the even one); the lists are walked
// behind one another, 64 runs a pass.  A piece with more runs than its list holds is parsed the single-wave way (probed again).
__device__ __forceinline__ void match_pair_strip(const uint8_t* ring, const uint32_t* tab, MatchRec* recs, uint16_t* ends, uint32_t rec_cap, uint32_t* strip,
                                                 const uint32_t* LAs, const uint8_t* LAe, uint32_t nA, const uint32_t* LBs, const uint8_t* LBe, uint32_t nB,
                                                 uint32_t* candS, uint8_t* candE, uint32_t si, uint32_t n, uint32_t cs, uint32_t tend,
                                                 uint32_t* table_free, uint32_t gen, uint32_t* inh_w) {      // table_free: told (gen) when this wave probes no more; inh_w: CM_INH of this tile
    const uint32_t lane = lane_id();
    const uint32_t ce = cs + 1024;                                   // (full tiles only: the strip lies inside the block)
    ParseState st; st.nseq = 0; st.enc = 0; st.ll0 = 0; st.cur = cs;
    if (n >= kMfLimit + 1 && cs <= n - kMfLimit) {
        uint32_t mlimit = n - kLastLiterals; if (mlimit > tend) mlimit = tend;
        const uint32_t last_q = n - kMfLimit;
        if (nA <= kCandCap && nB <= kCandCap) {
            CMP_STAT(10, 1); CMP_STAT(11, nA + nB); CMP_STAT(12, (nA + nB + 63) / 64);
            if (lane == 0) lds_store_release_local(table_free, gen);
            const uint32_t cs_off = src_ring_off(cs), total = nA + nB;
            for (uint32_t pl = 0; pl < total && st.nseq < rec_cap; pl += kCandPerPass) {
                const uint32_t i = pl + lane;
                const bool have = i < total, inB = i >= nA;
                uint32_t S = 0, E = 0;
                if (have) { S = inB ? LBs[i - nA] : LAs[i]; E = inB ? LBe[i - nA] : LAe[i]; }
                const uint32_t base = cs + (inB ? 512u : 0u);
                parse_pass(ring, recs, ends, rec_cap, cs, cs_off, mlimit, last_q, have, base + ((S & 255u) << 1), have ? S >> 8 : 1u, base + (E << 1) + kMinMatch, st, mlimit == tend ? inh_w : nullptr, gen);
            }
        } else {
            uint32_t ph[2];
            parse_piece<false>(ring, tab, recs, ends, rec_cap, candS, candE, cs, cs, cs + 512, mlimit, last_q, ph, 1u, st);
            wave_lds_fence_local();
            parse_piece<false>(ring, tab, recs, ends, rec_cap, candS, candE, cs, cs + 512, ce, mlimit, last_q, ph, 1u, st);
        }
    }
    strip_summary(strip, si, st, ce);
    if (lane == 0) lds_store_release_local(table_free, gen);                // (at the latest)
}

// ------------------------------------------------------------------------------ entry-point table (optional output)
// Every kHintEvery-th sequence of the block leaves a ROW: {where its token sits in the block, where its literals start in
// the source, how many sequences precede it} (lz4amd_params.h: lz4amd_hint_entry).  A decoder that is handed the table
// parses the block from all those rows at once, every lane the same number of sequences, instead of discovering the token
// chain (lz4_decompress_kernel.h: PARSER); the block itself is an ordinary LZ4 block.  The rows are written where the
// sequences are written out: a lane of the emit knows its sequence's place in the output and in the source.
struct HintOut { lz4amd_gdst table; uint32_t cap_rows, pre; uint32_t* over; uint32_t ord0, row0, k; };      // table == null: none wanted; ord0 / row0 / k: the tile's first ordinal, first row, log2 of its row distance
__device__ __forceinline__ void st_hint(lz4amd_gdst hints, uint32_t r, uint32_t tok, uint32_t out, uint32_t ord) { hint_store_row(hints, r, tok, out, ord); }
__device__ __forceinline__ void hint_row(const HintOut& H, bool on, uint32_t ord, uint32_t tok, uint32_t src_pos) {
    const uint32_t x = ord - H.ord0;
    if (on && (x & ((1u << H.k) - 1u)) == 0) {
        const uint32_t r = H.row0 + (x >> H.k);
        if (r < H.cap_rows) { if (r) st_hint(H.table, r, tok, src_pos - H.pre, ord); }      // (row 0 is written at the block's end: it also carries the number of rows)
        else *H.over = 1u;
    }
}

// wave copy of `len` literal bytes from block position sp to dst + d: out of the ring when the
// bytes are still there, else from HBM
__device__ __forceinline__ void copy_literals(lz4amd_gdst dst, uint32_t d, lz4amd_gsrc src, const uint8_t* ring,
                                              uint32_t sp, uint32_t len, uint32_t ring_lo) {
    const uint32_t lane = lane_id();
    // (8 bytes a lane: a tile that does not fit the staging buffer - it ends a literal run of many KB, the rule on data that hardly compresses - comes
    //  through here whole; byte stores made `datagen -P2` 1.8 times slower than -P60)
    const uint32_t n8 = len >> 3;
    if (sp >= ring_lo) {
        const uint32_t o = src_ring_off(sp);
        for (uint32_t k = lane; k < n8; k += 64) st_global8_raw(dst + d + 8 * k, ring_ld8(ring, ring_fwd(o, 8 * k)));
        for (uint32_t i = 8 * n8 + lane; i < len; i += 64) dst[d + i] = ring[ring_fwd(o, i)];
    } else {
        for (uint32_t k = lane; k < n8; k += 64) st_global8_raw(dst + d + 8 * k, ld_u64_g(src + sp + 8 * k));
        for (uint32_t i = 8 * n8 + lane; i < len; i += 64) dst[d + i] = src[sp + i];
    }
}

// ------------------------------------------------------------------------------ emit (one strip)
__device__ __forceinline__ void emit_strip(const uint8_t* ring, const MatchRec* recs, const uint32_t* strip,
                                           uint32_t w, lz4amd_gsrc src, lz4amd_gdst dst, uint32_t cs, uint32_t ring_lo, const HintOut* H = nullptr, uint32_t ord0 = 0) {      // H: the entry-point table's rows are written on the way (ord0: sequences of the block before this strip)
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
        if (H) hint_row(*H, have, ord0 + i, my_o, my_i);
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

// ------------------------------------------------------------------------------ emit straight to HBM, the plain way (one strip)
// For the rare tile whose encoded bytes do not fit the staging buffer (it ends a literal run of many KB): one sequence
// after the other, lane 0 writes the token / length / offset bytes, the wave copies the literals from the source in HBM.
// Small and slow on purpose: it must not cost the staged path registers.
__device__ __forceinline__ void emit_strip_plain(const MatchRec* recs, const uint32_t* strip, uint32_t w, lz4amd_gsrc src, lz4amd_gdst dst, uint32_t cs, const HintOut& H) {
    const uint32_t lane = lane_id();
    const uint32_t nk = strip[S_N * kCmpWaves + w];
    uint32_t ipos = cs - strip[S_CARRY * kCmpWaves + w];      // source position of the next sequence's literals (the first one's include the carried ones)
    uint32_t opos = strip[S_OUT * kCmpWaves + w];
#pragma nounroll
    for (uint32_t i = 0; i < nk; i++) {
        const MatchRec r = recs[i];
        const uint32_t mlm4 = r.mo >> 16, off = r.mo & 0xFFFFu;
        const uint32_t tl = i == 0 ? cs + (r.ll & 0xFFFFu) - ipos : (r.ll & 0xFFFFu);
        const uint32_t lit_d = opos + 1 + lit_hdr_ext(tl);
        if (H.table) hint_row(H, lane == 0, strip[S_ORD * kCmpWaves + w] + i, opos, ipos);
        if (lane == 0) {
            lz4amd_gdst p = dst + opos;
            *p++ = (uint8_t)(((tl >= 15 ? 15u : tl) << 4) | (mlm4 >= 15 ? 15u : mlm4));
            if (tl >= 15) p = put_len_ext(p, tl - 15);
            p = dst + lit_d + tl;
            p[0] = (uint8_t)off; p[1] = (uint8_t)(off >> 8); p += 2;
            if (mlm4 >= 15) p = put_len_ext(p, mlm4 - 15);
        }
#pragma nounroll
        for (uint32_t k = lane; k < tl; k += 64) dst[lit_d + k] = src[ipos + k];
        opos = lit_d + tl + 2 + (mlm4 >= 15 ? len_ext_bytes(mlm4 - 15) : 0u);
        ipos += tl + mlm4 + kMinMatch;
    }
}

// ------------------------------------------------------------------------------ emit into the staging buffer (one strip)
// The same sequences as emit_strip, composed in LDS: `stage` holds the tile's encoded bytes, stage[0] = output position
// dbase (the 16-byte chunk of the output the tile starts in).  Byte-granular LDS writes cost what dword writes cost
// (tools/exp/lds_prims.hip); the bytes then leave for HBM as whole aligned 16-byte chunks (flush_begin, flush_end).  Literals always
// come out of the source ring here: a staged tile's literals are at most kStageBytes old.
__device__ __forceinline__ void emit_strip_lds(const uint8_t* ring, const MatchRec* recs, const uint32_t* strip, uint32_t w,
                                               uint8_t* stage, uint32_t dbase, uint32_t cs, uint32_t* scr, const HintOut& H) {
    const uint32_t lane = lane_id();
    const uint32_t nk = strip[S_N * kCmpWaves + w];
    const uint32_t carry = strip[S_CARRY * kCmpWaves + w];
    const uint32_t cs_off = src_ring_off(cs);
    uint32_t ipos = 0;             // source position of the next sequence's own literals, from cs
    uint32_t opos = strip[S_OUT * kCmpWaves + w] - dbase;      // staging offset of the next sequence's token
    CMP_STAT(7, 1);
    for (uint32_t base = 0; base < nk; base += 64) {
        CMP_STAT(8, 1);
        const uint32_t i = base + lane;
        const bool have = i < nk;
        uint32_t ll = 0, mlm4 = 0, off = 0, extra = 0;
        if (have) { const MatchRec r = recs[i]; ll = r.ll & 0xFFFFu; off = r.mo & 0xFFFFu; mlm4 = r.mo >> 16; }      // (ll's upper half: parse_pass)
        if (i == 0) extra = carry;                       // literals inherited from earlier strips
        const uint32_t tl = ll + extra;
        const uint32_t lhdr = 1 + lit_hdr_ext(tl);       // token + the literal length's extension bytes
        const uint32_t e = have ? lhdr + tl + 2 + (mlm4 >= 15 ? len_ext_bytes(mlm4 - 15) : 0u) : 0u;
        const uint32_t adv = have ? ll + mlm4 + kMinMatch : 0;
        const uint32_t ea_incl = wave_incl_sum(e | (adv << 16));       // (one scan for both: a tile's encoded bytes and a strip's advance stay far below 64 K)
        const uint32_t e_incl = ea_incl & 0xFFFFu, a_incl = ea_incl >> 16;
        const uint32_t rel = ipos + a_incl - adv;        // my own literals start here (from cs); the carried ones lie before
        const uint32_t so = rel >= extra ? ring_fwd(cs_off, rel - extra) : ring_back(cs_off, extra - rel);
        const uint32_t my_o = opos + e_incl - e, lit_d = my_o + lhdr;
        if (H.table) hint_row(H, have, strip[S_ORD * kCmpWaves + w] + i, my_o + dbase, cs + rel - extra);
        // ---- the literal runs, in pieces of 8 bytes, lane = piece: pieces are numbered by a wave scan, a piece finds its
        //      sequence through a scatter of the sequences' first piece numbers + a running maximum (no walk over the
        //      sequences: a wave that hands itself one sequence after the other spends its time on LDS round trips)
        const uint32_t cnt = have ? (tl + 7) >> 3 : 0u;
        const uint32_t p_incl = wave_incl_sum(cnt), pbase = p_incl - cnt, npieces = wave_readlane(p_incl, 63);
        for (uint32_t W = 0; W < npieces; W += 64) {
            CMP_STAT(9, 1);
            scr[lane] = 0;
            wave_lds_fence_local();
            if (cnt && pbase < W + 64 && pbase + cnt > W) scr[pbase > W ? pbase - W : 0u] = lane + 1;
            wave_lds_fence_local();
            const uint32_t own = wave_incl_max(scr[lane]);
            wave_lds_fence_local();
            const uint32_t ol = own ? own - 1 : 0u;
            const uint32_t o_so = (uint32_t)__shfl((int)so, (int)ol), o_d = (uint32_t)__shfl((int)lit_d, (int)ol);
            const uint32_t o_tl = (uint32_t)__shfl((int)tl, (int)ol), o_pb = (uint32_t)__shfl((int)pbase, (int)ol);
            const uint32_t c8 = 8 * (W + lane - o_pb);
            const uint32_t nb = (own && W + lane < npieces) ? (o_tl - c8 < 8 ? o_tl - c8 : 8u) : 0u;
            if (nb) {
                // 8 source bytes out of three aligned dwords, cut to the piece's length, shifted to the destination's place in its
                // dwords and OR-ed in (the staging buffer is zero between tiles; the token / length / offset bytes of other lanes in
                // the same dwords are byte writes, which the OR of zero bytes leaves alone): no byte loop, no per-byte predicate
                const uint32_t sa = ring_fwd(o_so, c8);                    // (the ring's pad covers the bytes of a piece that crosses its end)
                const uint32_t* s32 = (const uint32_t*)(ring + (sa & ~3u));
                const uint32_t s0 = s32[0], s1 = s32[1], s2 = s32[2];
                uint32_t lo = align_bytes(s1, s0, sa & 3u), hi = align_bytes(s2, s1, sa & 3u);
                lo &= nb >= 4 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu << (8 * nb));
                hi &= nb >= 8 ? 0xFFFFFFFFu : (nb > 4 ? ~(0xFFFFFFFFu << (8 * (nb - 4))) : 0u);
                const uint32_t da = o_d + c8, sh = 8 * (da & 3u);
                const uint64_t v = ((uint64_t)lo | ((uint64_t)hi << 32)) << sh;
                uint32_t* d32 = (uint32_t*)(stage + (da & ~3u));
                atomicOr(&d32[0], (uint32_t)v);
                atomicOr(&d32[1], (uint32_t)(v >> 32));
                atomicOr(&d32[2], sh ? hi >> (32 - sh) : 0u);
            }
        }
        // ---- token, length bytes, offset
        if (have) {
            uint32_t o = my_o;
            const uint32_t tok_ll = tl >= 15 ? 15u : tl, tok_ml = mlm4 >= 15 ? 15u : mlm4;
            stage[o++] = (uint8_t)((tok_ll << 4) | tok_ml);
            if (tl >= 15) { uint32_t rest = tl - 15; while (rest >= 255) { stage[o++] = 255; rest -= 255; } stage[o++] = (uint8_t)rest; }
            o = lit_d + tl;
            stage[o] = (uint8_t)off; stage[o + 1] = (uint8_t)(off >> 8); o += 2;
            if (mlm4 >= 15) { uint32_t rest = mlm4 - 15; while (rest >= 255) { stage[o++] = 255; rest -= 255; } stage[o++] = (uint8_t)rest; }
        }
        opos += wave_readlane(e_incl, 63);
        ipos += wave_readlane(a_incl, 63);
    }
}

// ------------------------------------------------------------------------------ flush
// The staged tile leaves for HBM: a thread stores one ALIGNED 16-byte chunk, number ci of the staging buffer (the output's own
// 16-byte grid: a0 is dst's misalignment) - every thread its own after a small tile, 64 chunks a wave from a queue after a full one.  The chunk the tile starts in begins with the last bytes of the tile before: they wait in the
// carry chunk (two of them, by tile parity) and leave now; the bytes behind the tile's last whole chunk wait in turn.
// cfrom: first byte of the carry chunk that is pending (bytes below it left already, or lie before dst).
struct FlushCtx { U32x4 v; uint32_t dbase, h, nfull, r, cfrom, direct, pp; };
static_assert(kStageBytes / 16 <= kCmpThreads && kStageBytes % 1024 == 0, "one staged chunk per thread");
// first half: the tile's numbers and my chunk of the staging buffer into registers (the table inserts run while they arrive)
__device__ __forceinline__ FlushCtx flush_begin(char* smem, uint32_t pp, uint32_t a0, uint32_t ci) {
    const uint32_t* T = (const uint32_t*)(smem + kCOffMisc) + CM_TILE + 4 * pp;
    FlushCtx f;
    const uint32_t out0 = T[T_OUT0], out1 = T[T_OUT1];
    f.direct = T[T_DIRECT]; f.cfrom = T[T_CFROM]; f.pp = pp;
    const uint32_t V0 = out0 + a0, V1 = out1 + a0;
    f.dbase = V0 & ~15u; f.h = V0 - f.dbase;
    f.nfull = (V1 - f.dbase) >> 4; f.r = (V1 - f.dbase) & 15u;
    f.v = *(const U32x4*)(smem + kCOffStage + 16 * (ci < kStageBytes / 16 ? ci : 0u));       // (threads behind the tile's last chunk read bytes nobody uses)
    if (16 * ci < kStageBytes) { U32x4 z; z[0] = z[1] = z[2] = z[3] = 0; *(U32x4*)(smem + kCOffStage + 16 * ci) = z; }   // the next tile ORs its literals into zeros
    return f;
}
__device__ __forceinline__ void flush_end(char* smem, const FlushCtx& f, lz4amd_gdst dst, uint32_t a0, uint32_t ci) {
    const uint32_t tid = ci;
    uint32_t* Tn = (uint32_t*)(smem + kCOffMisc) + CM_TILE + 4 * (f.pp ^ 1);
    const uint8_t* C = (const uint8_t*)(smem + kCOffCarry) + 16 * f.pp;
    uint8_t* Cn = (uint8_t*)(smem + kCOffCarry) + 16 * (f.pp ^ 1);
    if (f.direct) {
        // the tile went to HBM byte by byte (emit_strip): only the pending bytes before it are left to write
        if (tid >= f.cfrom && tid < f.h) dst[f.dbase - a0 + tid] = C[tid];
        if (tid == 0) Tn[T_CFROM] = (f.dbase + 16 * f.nfull + f.r) & 15u;
        return;
    }
    if (tid < f.nfull || (tid == f.nfull && f.r)) {
        U32x4 v = f.v;
        if (tid == 0 && f.h) {
            const U32x4 c = *(const U32x4*)C;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const int32_t rr = (int32_t)f.h - 4 * (int32_t)k;
                const uint32_t m = rr >= 4 ? 0xFFFFFFFFu : (rr <= 0 ? 0u : ((1u << (8 * rr)) - 1u));
                v[k] = (c[k] & m) | (v[k] & ~m);
            }
        }
        if (tid < f.nfull) {
            if (tid == 0 && f.cfrom) {
#pragma nounroll
                for (uint32_t k = f.cfrom; k < 16; k++) dst[f.dbase - a0 + k] = (uint8_t)(v[k >> 2] >> (8 * (k & 3)));
            } else st_global16(dst + (f.dbase - a0 + 16 * tid), v);
        } else *(U32x4*)Cn = v;
    }
    if (tid == 0) Tn[T_CFROM] = f.nfull ? 0u : f.cfrom;
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
__device__ __forceinline__ void resolve_overruns(uint32_t* strip, MatchRec* recs_tile, const uint16_t* ends_tile, uint32_t rps,
                                                 uint32_t nstrips, uint32_t g0, uint32_t t0, uint32_t strip_len, uint32_t t1, uint32_t n) {      // rps: record slots per strip
    const uint32_t lane = lane_id();
    const bool mine = lane < nstrips;
    const uint32_t cs = strip_lo(g0, t0, lane, strip_len);
    uint32_t ce = g0 + (lane + 1) * strip_len; if (ce > t1) ce = t1;
    const uint32_t nk = mine ? strip[S_N * kCmpWaves + lane] : 0, own_end = mine ? strip[S_END * kCmpWaves + lane] : 0;
    MatchRec* rk = recs_tile + lane * rps;
    const uint16_t* ek = ends_tile + lane * rps;
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
                const uint32_t cs_k = strip_lo(g0, t0, k, strip_len), Pk = cover > cs_k ? cover : cs_k;
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
            // first record whose match ends behind P: nearly always the first or the second one (an overrun covers a few
            // bytes of the next strip), or none (a long match covers the whole strip): look there before searching
            uint32_t f;
            if (nk == 0 || cs + ek[nk - 1] <= P) f = nk;
            else if (cs + ek[0] > P) f = 0;
            else {
                uint32_t lo = 1, hi = nk - 1;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cs + ek[mid] > P) hi = mid; else lo = mid + 1; }
                f = lo;
            }
            first = nk; nk2 = 0; enc = 0; ll0 = 0; tail = ce > P ? ce - P : 0;
            if (f < nk) {
                MatchRec r = rk[f];
                const uint32_t enc_before = r.ll >> 16;              // encoded bytes of the records that go
                r.ll &= 0xFFFFu;
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
                    enc = enc_total - enc_before - old_enc + enc_size(r.ll, r.mo >> 16);
                } else if (f + 1 < nk) {
                    // its bytes [P, ef) are literals of the next record
                    MatchRec r2 = rk[f + 1];
                    r2.ll &= 0xFFFFu;
                    const uint32_t old2 = enc_size(r2.ll, r2.mo >> 16);
                    r2.ll += ef - P;
                    rk[f + 1] = r2;
                    first = f + 1; nk2 = nk - f - 1; ll0 = r2.ll;
                    enc = enc_total - (enc_before + old_enc) - old2 + enc_size(r2.ll, r2.mo >> 16);
                }
                if (nk2) { const uint32_t e_last = cs + ek[nk - 1]; tail = e_last < ce ? ce - e_last : 0; }
            }
        }
        strip[S_N * kCmpWaves + lane] = nk2; strip[S_ENC * kCmpWaves + lane] = enc; strip[S_LL0 * kCmpWaves + lane] = ll0;
        strip[S_TAIL * kCmpWaves + lane] = tail; strip[S_FIRST * kCmpWaves + lane] = first; strip[S_P * kCmpWaves + lane] = P;
    }
}

struct StripTotals { uint32_t out, carry, fail, seqs; };
__device__ __forceinline__ StripTotals strip_offsets(uint32_t* strip, uint32_t nstrips, uint32_t out0, uint32_t carry0,
                                                     uint32_t fail, uint32_t cap, uint32_t seq0 = 0, bool with_ordinals = false) {
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
    const uint32_t n_incl = wave_incl_sum(nk);
    if (mine) { strip[S_OUT * kCmpWaves + lane] = out0 + o_incl - sz; strip[S_CARRY * kCmpWaves + lane] = my_carry; }
    if (mine && with_ordinals) strip[S_ORD * kCmpWaves + lane] = seq0 + n_incl - nk;      // (the HC kernel's strip array has no such row)
    StripTotals r;
    r.out = fail ? out0 : out0 + total;
    r.carry = ne ? t_total - wave_readlane(t_excl, last) : carry0 + t_total;
    r.fail = fail;
    r.seqs = seq0 + wave_readlane(n_incl, 63);
    return r;
}

// a control word every lane of the wave agrees on
__device__ __forceinline__ uint32_t uload_cm(const uint32_t* w) { return __builtin_amdgcn_readfirstlane(lds_load_acquire_local(w)); }

// ------------------------------------------------------------------------------ one tile settled / written (helpers of one block)
// wave 0: tile (parity pp) gets its output offsets; whether it is composed in LDS or - when its encoded bytes do not fit
// the staging buffer, i.e. when it ends a literal run of more than a few KB - written to HBM directly
__device__ __forceinline__ void settle_tile(char* smem, uint32_t pp, uint32_t nstrips, uint32_t g0, uint32_t t0, uint32_t strip_len, uint32_t t1,
                                            uint32_t n, uint32_t cap, uint32_t a0) {
    const uint32_t rps = strip_len == 1024 ? kRecsPerPair : kRecsPerStrip;
    uint32_t* misc = (uint32_t*)(smem + kCOffMisc);
    uint32_t* strip_p = (uint32_t*)(smem + kCOffStrip) + pp * kStripFields * kCmpWaves;
#ifdef LZ4AMD_PROF_TILE
    const uint64_t ts0 = clock_ticks();
#endif
    resolve_overruns(strip_p, (MatchRec*)(smem + kCOffRecs) + pp * kRecsPerTile, (const uint16_t*)(smem + kCOffEnds) + pp * kRecsPerTile, rps, nstrips, g0, t0, strip_len, t1, n);
    wave_lds_fence_local();
#ifdef LZ4AMD_PROF_TILE
    const uint64_t ts1 = clock_ticks();
#endif
    const uint32_t out0 = misc[CM_OUT];
    const uint32_t seq0 = misc[CM_SEQS];
    const StripTotals t = strip_offsets(strip_p, nstrips, out0, misc[CM_CARRY], misc[CM_FAIL], cap, seq0, true);
    if (lane_id() == 0) {
        misc[CM_OUT] = t.out; misc[CM_CARRY] = t.carry; misc[CM_FAIL] = t.fail; misc[CM_SEQS] = t.seqs;
        // the tile's rows of the entry-point table: one per 2^k sequences, k from the tile's own number of sequences - about one
        // row per 512 bytes of source, never more than 8 sequences apart - 16 where that still is under 512 bytes, or small blocks of such data
        // would not fit their rows in the table's room (a lane of the decoder's parser walks a row's sequences
        // one by one).  (The tile before is no guide: a block's first tile, parsed against an empty table, has none.)
        const uint32_t lg = 31u - (uint32_t)__clz((int)(t1 - t0 ? t1 - t0 : 1u));
        const uint32_t want = (t.seqs - seq0) >> (lg > 9 ? lg - 9 : 0u);              // sequences per 512 bytes
        const uint32_t k = want >= LZ4AMD_HINT_EVERY_MAX ? LZ4AMD_HINT_EVERY_LOG2 : want >= 8 ? 3u : want >= 4 ? 2u : want >= 2 ? 1u : 0u;
        uint32_t* HT = misc + CM_HTILE + 4 * pp;
        HT[0] = seq0; HT[1] = misc[CM_ROWS]; HT[2] = k;
        misc[CM_ROWS] += (t.seqs - seq0 + (1u << k) - 1) >> k;
        uint32_t* T = misc + CM_TILE + 4 * pp;
        T[T_OUT0] = out0; T[T_OUT1] = t.out;
        T[T_DIRECT] = (t.out + a0) - ((out0 + a0) & ~15u) > kStageBytes - 16 ? 1u : 0u;
    }
    wave_lds_fence_local();
#ifdef LZ4AMD_PROF_TILE
    { const uint64_t ts2 = clock_ticks(); if (lane_id() == 0) { ((uint64_t*)(smem + kCOffMisc))[12] += ts1 - ts0; ((uint64_t*)(smem + kCOffMisc))[13] += ts2 - ts1; } }
#endif
}
// every wave: its strip of the settled tile (parity pp), into the staging buffer or straight to HBM
// (w: the strip; sw: the wave that does it - any wave may, the strip's place in the output was fixed when the tile was settled)
__device__ __forceinline__ void emit_tile_strip(char* smem, uint32_t pp, uint32_t w, uint32_t sw, uint32_t rps, lz4amd_gsrc src, lz4amd_gdst dst, uint32_t a0, uint32_t ring_lo, const HintOut& H0) {      // rps: record slots per strip of that tile
    const uint32_t* misc = (const uint32_t*)(smem + kCOffMisc);
    const uint32_t* strip_p = (const uint32_t*)(smem + kCOffStrip) + pp * kStripFields * kCmpWaves;
    if (misc[CM_FAIL] || !strip_p[S_N * kCmpWaves + w]) return;
    const uint8_t* ring = (const uint8_t*)(smem + kCOffRing);
    const MatchRec* recs_w = (const MatchRec*)(smem + kCOffRecs) + pp * kRecsPerTile + w * rps + strip_p[S_FIRST * kCmpWaves + w];
    const uint32_t* T = misc + CM_TILE + 4 * pp;
    HintOut H = H0;
    if (H.table) { const uint32_t* HT = misc + CM_HTILE + 4 * pp; H.ord0 = HT[0]; H.row0 = HT[1]; H.k = HT[2]; }
    if (T[T_DIRECT]) emit_strip_plain(recs_w, strip_p, w, src, dst, strip_p[S_P * kCmpWaves + w], H);
    else emit_strip_lds(ring, recs_w, strip_p, w, (uint8_t*)(smem + kCOffStage), ((T[T_OUT0] + a0) & ~15u) - a0, strip_p[S_P * kCmpWaves + w],
                        (uint32_t*)(smem + kCOffScr) + sw * 64, H);    // (scratch: the executing wave's own; stage[0] = the chunk's first byte; wraps for the first chunk of an unaligned dst)
}

// the 8 positions q0 .. q0 + 7 (q0: a multiple of 8) into the table - those that may start a match (below t1, up to last_q) -,
// hashed out of four aligned dwords; have_h: the even ones' table indices are in probe_h (the thread probed them)
__device__ __forceinline__ void insert_unit(const uint8_t* ring, uint32_t* tab, uint32_t q0, uint32_t t1, uint32_t last_q, bool small, bool have_h, const uint32_t (&probe_h)[2]) {
    if (q0 < t1 && q0 <= last_q) {
        const uint32_t o = src_ring_off(q0);                // multiple of 8: o + 16 <= ring + pad
        const uint32_t* a = (const uint32_t*)(ring + o);
        uint32_t dw[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) dw[i] = a[i];
        uint32_t h[8];
        // (the position's first four bytes by one byte alignment - none for the aligned ones -, its fifth byte by one bit-field extract)
#define LZ4AMD_INS_HASH(i) hash_pos32((i) & 3 ? align_bytes(dw[(i) / 4 + 1], dw[(i) / 4], (i) & 3) : dw[(i) / 4], dw[((i) + 4) / 4] >> (8 * (((i) + 4) & 3)), small)
        h[1] = LZ4AMD_INS_HASH(1); h[3] = LZ4AMD_INS_HASH(3); h[5] = LZ4AMD_INS_HASH(5); h[7] = LZ4AMD_INS_HASH(7);
        if (have_h) {                                           // computed when the strip was probed (wave-uniform)
            h[0] = probe_h[0] & 0xFFFFu; h[2] = probe_h[0] >> 16; h[4] = probe_h[1] & 0xFFFFu; h[6] = probe_h[1] >> 16;
        } else {
            h[0] = LZ4AMD_INS_HASH(0); h[2] = LZ4AMD_INS_HASH(2); h[4] = LZ4AMD_INS_HASH(4); h[6] = LZ4AMD_INS_HASH(6);
        }
#undef LZ4AMD_INS_HASH
        if (q0 + 7 < t1 && q0 + 7 <= last_q) {                  // every thread but the ones at a block's very end: no per-position test
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) atomicMax(&tab[h[i]], q0 + i);
        } else {
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) if (q0 + i < t1 && q0 + i <= last_q) atomicMax(&tab[h[i]], q0 + i);
        }
    }
}

// ------------------------------------------------------------------------------ one block
__device__ __forceinline__ void compress_one_block(const CompBatch& P, uint32_t b, char* smem) {
    const uint32_t tid = threadIdx.x, w = wave_id();
    uint32_t* misc = (uint32_t*)(smem + kCOffMisc);
    uint32_t* strip = (uint32_t*)(smem + kCOffStrip);
    uint32_t* tab = (uint32_t*)(smem + kCOffTab);
    const uint32_t sw = w;
    MatchRec* recs = (MatchRec*)(smem + kCOffRecs);                            // + parity * kRecsPerTile + strip * its slots
    uint16_t* ends = (uint16_t*)(smem + kCOffEnds);
    uint32_t* candS = (uint32_t*)(smem + kCOffCandS) + sw * kCandCap;
    uint8_t* candE = (uint8_t*)(smem + kCOffCandE) + sw * kCandCap;
    uint32_t* pairw = (uint32_t*)(smem + kCOffPair);
    uint8_t* ring = (uint8_t*)(smem + kCOffRing);

    // history (linked blocks, lz4io.c:741-744 / LZ4_compress_fast_continue in prefix mode lz4.c:1707): the
    // `pre` bytes right before the block are parsed into the table but not emitted (all but up to 15 of them: the
    // block starts on the 16-byte grid of the positions).
    uint32_t pre = P.prefix ? (uint32_t)P.prefix[b] : 0u;
    if (pre > kMaxDistance + 1) pre = kMaxDistance + 1;
    pre &= ~15u;
    const lz4amd_gsrc src = LZ4AMD_TO_GSRC(P.src[b]) - pre;          // position 0 = start of the history
    const lz4amd_gdst dst = LZ4AMD_TO_GDST(P.dst[b]);
    const int32_t n_i = P.src_size[b];
    const int32_t cap_i = P.dst_cap[b];
    if (n_i < 0 || (uint32_t)n_i > kMaxInput || cap_i <= 0 || P.dst[b] == nullptr || (P.src[b] == nullptr && n_i != 0)) {
        if (tid == 0) { P.result[b] = 0; if (P.hints) *(uint32_t*)(P.hints + (uint64_t)b * P.hint_stride) = 0; }      // lz4.c:1360
        return;
    }
    if (n_i == 0) { if (tid == 0) { dst[0] = 0; P.result[b] = 1; if (P.hints) *(uint32_t*)(P.hints + (uint64_t)b * P.hint_stride) = 0; } return; }      // lz4.c:1361-1371 (no table for an empty block)
    const uint32_t n = (uint32_t)n_i + pre, cap = (uint32_t)cap_i;
    const bool small = (uint32_t)n_i < kSmallBlockLimit;       // (the block's own size: a small block is probed at every position whatever history precedes it)
    const bool stride4 = !small && P.acceleration >= LZ4AMD_STRIDE4_FROM;      // LZ4_compress_fast's speed / ratio knob: every fourth position is probed instead of every second
    const uint32_t a0 = (uint32_t)((uintptr_t)P.dst[b] & 15u);        // dst's place on HBM's 16-byte grid
    const lz4amd_gdst hints = P.hints ? LZ4AMD_TO_GDST(P.hints + (uint64_t)b * P.hint_stride) : (lz4amd_gdst)nullptr;      // optional entry-point table
    HintOut H; H.table = hints; H.cap_rows = LZ4AMD_HINT_CAP_ROWS(P.hint_stride); H.pre = pre; H.over = &misc[CM_HOVER]; H.ord0 = H.row0 = H.k = 0;

    for (uint32_t i = tid; i < (1u << kHashBits); i += kCmpThreads) tab[i] = 0;
    if (16 * tid < kStageBytes) { U32x4 z; z[0] = z[1] = z[2] = z[3] = 0; *(U32x4*)(smem + kCOffStage + 16 * tid) = z; }
    if (tid == 0) {
        misc[CM_OUT] = 0; misc[CM_CARRY] = 0; misc[CM_FAIL] = 0; misc[CM_READY] = 0; misc[CM_SEQS] = 0; misc[CM_INH] = 0; misc[CM_INH + 1] = 0;
        for (uint32_t i = CM_EMITQ; i < CM_FLUSHQ + 2; i++) misc[i] = 0; misc[CM_HOVER] = 0; misc[CM_ROWS] = 0;
        for (uint32_t i = 0; i < 2 * kCmpWaves; i++) pairw[i] = 0;
#ifdef LZ4AMD_PROF_TILE
        for (uint32_t i = 24; i < 30; i++) misc[i] = 0;
#endif
        misc[CM_TILE + T_CFROM] = a0; misc[CM_TILE + 4 + T_CFROM] = a0;     // nothing of the first chunk is pending: the bytes before dst are not ours
    }
    // first tile straight into the ring; later tiles are prefetched one tile ahead
    uint32_t t0 = 0, tile_len, strip_len;
    tile_geometry(pre ? kTileMax * 4 : 0, small, tile_len, strip_len);      // (the history goes in in the largest tiles; the block itself starts with small ones)
    uint32_t loaded = 0;                                  // ring holds [.., loaded)
    uint64_t* prof = (LZ4AMD_CMP_PROF && P.prof) ? P.prof + (uint64_t)blockIdx.x * 8 : nullptr;      // (a developer build's stamps: the product kernel carries none of that code)
    uint64_t tp[5] = {0, 0, 0, 0, 0}, tq = 0;
#ifdef LZ4AMD_PROF_ROLES
    // developer build (tools/prof_roles.py): cycles of wave LZ4AMD_PROF_ROLES in the full tiles where its role is LZ4AMD_PROF_ROLES_SEL (0 measuring, 1 writing, 2 either):
    // (settle,) probe + list | partner wait, measure | settle wait, emit | wait for the insert gate | insert | wait for the flush gate | flush | wait at the barrier
#ifndef LZ4AMD_PROF_ROLES_SEL
#define LZ4AMD_PROF_ROLES_SEL 2
#endif
    uint64_t rt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t rq = 0; bool ron = false;
#endif
    if (prof) tq = clock_ticks();
    {
        uint32_t hi = tile_len + 16; if (hi > n) hi = n;
        for (uint32_t Pp = 16 * tid; Pp < hi; Pp += 16 * kCmpThreads) ring_commit16(ring, Pp, load_src16(src, n, Pp));
        loaded = (hi + 15) & ~15u;
    }
    // One barrier per full tile (at its top), two per small one.  A full tile k: one writing wave settles tile k-1 (overrunning matches, then the
    // strips' sizes into output offsets: ~4 K cycles of one wave's dependent work) and says so in CM_READY; fifteen waves probe and list the
    // tile's sixteen pieces; one wave of every pair measures and selects; whoever is free composes tile k-1's strips in the staging buffer once
    // CM_READY covers it, inserts tile k's pieces into the table once all are probed, and stores tile k-1's bytes once all its strips are
    // composed - all handed out from queues in LDS (two sets, by tile parity).  A small tile has a second barrier, behind which everybody
    // inserts its own positions and stores one 16-byte chunk.  Records and strip summaries are double buffered.
    U32x4 pf; pf[0] = pf[1] = pf[2] = pf[3] = 0;            // the granule a thread fetches a tile ahead (declared out here: cleared at the loop's top, the compiler would wait for the
                                                            //  memory counter there - it does not know whether the load of the trip before was waited for -, i.e. for the
                                                            //  acknowledgements of the stores of the tile before as well)
    uint32_t par = 0;                                       // buffer parity of tile k
    uint32_t prev_t0 = 0, prev_g0 = 0, prev_t1 = 0, prev_strip_len = 0, prev_nstrips = 0, prev_rps = kRecsPerStrip;      // tile k-1, still to be emitted (rps: record slots per strip)
    uint32_t tiles_parsed = 0;                              // tiles whose strips were matched so far (CM_READY counts up to it)
    while (t0 < n) {
        uint32_t t1 = t0 + tile_len; if (t1 > n || t1 < t0) t1 = n;
        if (t0 < pre && t1 > pre) t1 = pre;                    // the block's first tile starts where the history ends
        uint32_t* strip_k = strip + par * kStripFields * kCmpWaves;
        MatchRec* recs_k = recs + par * kRecsPerTile;
        uint16_t* ends_k = ends + par * kRecsPerTile;
        // -- prefetch: the next tile's bytes (one 16-byte granule per thread, committed after the parse)
        uint32_t nt_len, nt_strip;
        tile_geometry(t1 >= pre ? t1 - pre : kTileMax * 4, small, nt_len, nt_strip);
        // (the ring's commit points assume that `loaded` is never more than 16 bytes ahead of the next tile's end: the history's last tile is cut
        //  at `pre`, which need not be a multiple of the tile - what was fetched beyond the cut counts, nothing is fetched twice)
        uint32_t pf_hi = t1 + nt_len + 16; if (pf_hi > n || pf_hi < t1) pf_hi = n;     // stays 16 bytes ahead of the tile
        if (pf_hi < loaded) pf_hi = loaded;
        // (a tile is at most 512 granules: the upper eight waves fetch them - not the wave that settles the tile before: wherever the
        //  compiler waits for the granule, that wave would wait at the head of the chain everybody else waits for)
        const uint32_t Pp = loaded + 16 * (tid ^ 512u);
        if (Pp < pf_hi) pf = load_src16(src, n, Pp);           // nt_len <= 8 * kCmpThreads
        __syncthreads();                                       // ring, table and tile k-1's records ready
        if (tid == 0) { misc[CM_EMITQ + (par ^ 1)] = 0; misc[CM_INSQ + (par ^ 1)] = 0; misc[CM_EMITDONE + (par ^ 1)] = 0; misc[CM_FLUSHQ + (par ^ 1)] = 0; }      // (the tile before's set: nobody looks at it any more)
        if (prof) { const uint64_t t = clock_ticks(); tp[0] += t - tq; tq = t; }
        const bool parse = t0 >= pre;
        const uint32_t g0 = strip_origin(t0, tile_len, strip_len);
        const uint32_t nstrips = parse ? (t1 - g0 + strip_len - 1) >> (31 - __clz((int)strip_len)) : 0;      // (strip lengths are powers of two)
        const uint32_t ring_lo = loaded > kSrcRing ? loaded - kSrcRing : 0;
        // a full tile of a big block: 16 pieces of 512 bytes, probed by one wave each, 8 strips of 1 KB, measured by one wave of a pair each
        const bool paired = parse && !small && !stride4 && strip_len == 512 && t1 - t0 == 8192;
        uint32_t probe_h[2] = {0, 0}; bool probe_h_valid = false;
#ifdef LZ4AMD_PROF_ROLES
        { const uint64_t t_ = clock_ticks(); if (ron) rt[7] += t_ - rq; rq = t_; }
        ron = paired && (LZ4AMD_PROF_ROLES_SEL == 2 || ((w ^ (w >> 2) ^ tiles_parsed) & 1u) == LZ4AMD_PROF_ROLES_SEL);
#define RSTAMP(k) do { const uint64_t t_ = clock_ticks(); if (ron) rt[k] += t_ - rq; rq = t_; } while (0)
#else
#define RSTAMP(k) do {} while (0)
#endif
        if (paired) {
            // -- A0: one writing wave settles tile k-1, before anything else: the other writing waves wait for it.  It does not probe: its
            //    partner - which measures the pair's strip anyway - probes both pieces (no list comes late, and the settling wave is free to
            //    write out as soon as it has settled: eight strips, eight writing waves)
            const uint32_t settle_w = (tiles_parsed & 1u) ? 5u : 1u;      // (an odd wave - its piece is its pair's second - whose role this tile is to write)
            if (w == settle_w && prev_nstrips) {
#if LZ4AMD_CMP_PRIO & 1
                wave_priority(3);
#endif
                settle_tile(smem, par ^ 1, prev_nstrips, prev_g0, prev_t0, prev_strip_len, prev_t1, n, cap, a0);
                if (lane_id() == 0) lds_store_release_local(&misc[CM_READY], tiles_parsed);
#if LZ4AMD_CMP_PRIO & 1
                wave_priority(0);
#endif
            }
            // -- A1: probe and list my piece, tell my partner
            const uint32_t last_q = n - kMfLimit;
            const uint32_t pcs = t0 + 512 * w, gen = (tiles_parsed & 0x7FFFu) + 1u;
            const uint32_t q_hi = pcs + 511 < last_q ? pcs + 511 : last_q;
            uint32_t nw = 0, nw2 = 0;
            // the distance of the match that ran into the end of the tile before (CM_INH of that tile's parity: written before this tile's barrier, by nobody now)
            const uint32_t inh_key = misc[CM_INH + (par ^ 1u)];
            const uint32_t inh_d = (tiles_parsed && (inh_key >> 16) == (((tiles_parsed - 1u) & 0x7FFFu) + 1u)) ? (inh_key & 0xFFFFu) : 0u;
            if (w != settle_w) {
                if (pcs <= last_q) { nw = probe_list<false>(ring, tab, candS, candE, pcs, src_ring_off(pcs), q_hi, 0u, 1u, probe_h, t0, inh_d); probe_h_valid = true; }
                wave_lds_fence_local();
                if (lane_id() == 0) lds_store_release_local(&pairw[w], (gen << 16) | nw);
                if (w == (settle_w ^ 1u)) {                        // (the settling wave's piece, into its list)
                    const uint32_t pcs2 = pcs + 512, q_hi2 = pcs2 + 511 < last_q ? pcs2 + 511 : last_q;
                    uint32_t ph2[2];
                    if (pcs2 <= last_q) nw2 = probe_list<false>(ring, tab, (uint32_t*)(smem + kCOffCandS) + settle_w * kCandCap, (uint8_t*)(smem + kCOffCandE) + settle_w * kCandCap,
                                                                pcs2, src_ring_off(pcs2), q_hi2, 0u, 1u, ph2, t0, inh_d);
                    wave_lds_fence_local();
                    if (lane_id() == 0) lds_store_release_local(&pairw[settle_w], (gen << 16) | nw2);
                }
            }
#if LZ4AMD_CMP_EARLY_COMMIT
            // the next tile's granules (fetched at the tile's top by the upper waves: they have arrived) go into the ring now: its slots hold
            // bytes more than a window below this tile - nobody reads them any more
            touch_load16(pf);       // (every wave, fetching or not: or the compiler, which does not know at the loop's top whether the granule's load was waited for,
                                    //  waits there before it clears these registers - for the memory counter, i.e. for the acknowledgements of the stores of the tile
                                    //  before as well, ~3 K cycles with the whole workgroup behind it at the barrier)
            if (Pp < pf_hi) ring_commit16(ring, Pp, pf);
#endif
            RSTAMP(0);
            // roles: of the two waves of a pair one measures, one writes out; they swap every tile, and a SIMD (waves w, w + 4, w + 8, w + 12) has two of each
            const uint32_t role = (w ^ (w >> 2) ^ tiles_parsed) & 1u;
            if (role == 0) {
#if LZ4AMD_CMP_PRIO & 2
                wave_priority(2);
#endif
                // -- A1': the pair's 1 KB, from both lists (the even wave's piece comes first)
                const uint32_t pw = w ^ 1u;
                uint32_t pv = nw2;
                if (pw != settle_w) while (((pv = uload_cm(&pairw[pw])) >> 16) != gen) spin_pause();
                const uint32_t np = pv & 0xFFFFu;
                const uint32_t* pS = (const uint32_t*)(smem + kCOffCandS) + pw * kCandCap;
                const uint8_t* pE = (const uint8_t*)(smem + kCOffCandE) + pw * kCandCap;
                const uint32_t si = w >> 1;
                if (w & 1u) match_pair_strip(ring, tab, recs_k + si * kRecsPerPair, ends_k + si * kRecsPerPair, kRecsPerPair, strip_k,
                                             pS, pE, np, candS, candE, nw, candS, candE, si, n, t0 + 1024 * si, t1, &pairw[kCmpWaves + si], gen, &misc[CM_INH + par]);
                else match_pair_strip(ring, tab, recs_k + si * kRecsPerPair, ends_k + si * kRecsPerPair, kRecsPerPair, strip_k,
                                      candS, candE, nw, pS, pE, np, candS, candE, si, n, t0 + 1024 * si, t1, &pairw[kCmpWaves + si], gen, &misc[CM_INH + par]);
#if LZ4AMD_CMP_PRIO & 2
                wave_priority(0);
#endif
                RSTAMP(1);
            }
        } else {
            // -- A0: one wave settles tile k-1 first
            // (a tile of eight strips or fewer - a small block's, a block's first ones - leaves the upper waves without a strip: one of
            //  them settles, and they write out tile k-1 while the lower ones match)
            if (w == (LZ4AMD_CMP_IDLE_SETTLE && nstrips <= 8 ? 8u : (uint32_t)kSettleWave) && prev_nstrips) {
                settle_tile(smem, par ^ 1, prev_nstrips, prev_g0, prev_t0, prev_strip_len, prev_t1, n, cap, a0);
                if (lane_id() == 0) lds_store_release_local(&misc[CM_READY], tiles_parsed);
            }
            // -- A1: match, one wave per strip (tiles of the history are only inserted into the table)
            if (w < nstrips) {
                const uint32_t cs = strip_lo(g0, t0, w, strip_len);
                uint32_t ce = g0 + (w + 1) * strip_len; if (ce > t1) ce = t1;
                if (small) match_strip<true>(ring, tab, recs_k + w * kRecsPerStrip, ends_k + w * kRecsPerStrip, kRecsPerStrip, strip_k, candS, candE, w, n, cs, ce, t1, probe_h, 0u);
                else match_strip<false>(ring, tab, recs_k + w * kRecsPerStrip, ends_k + w * kRecsPerStrip, kRecsPerStrip, strip_k, candS, candE, w, n, cs, ce, t1, probe_h, stride4 ? 2u : 1u);
                // the lane probed positions cs + 8 * lane + {0, 2, 4, 6}: with 512-byte strips those are this thread's insert positions
                probe_h_valid = !small && !stride4 && strip_len == 512 && n >= kMfLimit + 1 && cs <= n - kMfLimit;
            }
            touch_load16(pf);       // (see above: the code from here on is shared with the full tiles)
        }
        if (prof) { const uint64_t t = clock_ticks(); tp[1] += t - tq; tq = t; }
        // -- A2: write out tile k-1 (into the staging buffer)
        //    The strips are handed out from a counter to whichever wave is free: the waves do not finish their matching together
        //    (the wave that settled the tile starts late, and the arbiter serves a SIMD's low wave slots first), and the barrier
        //    below waits for the last one.
        if (prev_nstrips) {
            while (uload_cm(&misc[CM_READY]) < tiles_parsed) CMP_WAIT_PAUSE();
            for (;;) {
                uint32_t sx = 0;
                if (lane_id() == 0) sx = atomicAdd(&misc[CM_EMITQ + par], 1u);
                sx = __builtin_amdgcn_readfirstlane(sx);
                if (sx >= prev_nstrips) break;
#if LZ4AMD_CMP_EMIT_PRIO
                wave_priority(LZ4AMD_CMP_EMIT_PRIO);                     // (writing out: above the inserts and stores, below the measuring waves - datagen -P20 3.30 -> 3.21 ms per GiB, -P60 / -P90 unchanged)
#endif
                emit_tile_strip(smem, par ^ 1, sx, w, prev_rps, src, dst, a0, ring_lo, H);
#if LZ4AMD_CMP_EMIT_PRIO
                wave_priority(0);
#endif
                wave_lds_fence_local();
                lds_or_release_local(&misc[CM_EMITDONE + par], 1u << sx);          // (every lane the same bit: a store by lane 0 alone, in this loop, hung the kernel on the device)
            }
        }
        RSTAMP(2);
        if (paired) {
            // -- A3: the tile goes into the table, piece by piece, by whichever wave is free (the writing waves mostly: the measuring ones have the
            //    longer way to the barrier) - once all sixteen pieces were probed and no measuring wave has to probe one again: the table is
            //    frozen while it is read
            const uint32_t gen = (tiles_parsed & 0x7FFFu) + 1u;
            for (;;) {
                const uint32_t l = lane_id();
                const uint32_t f = l < kCmpWaves ? lds_load_acquire_local(&pairw[l]) >> 16 : l < kCmpWaves + 8 ? lds_load_acquire_local(&pairw[l]) : gen;
                if (__all(f == gen)) break;
                CMP_WAIT_PAUSE();
            }
            RSTAMP(3);
            for (;;) {
                uint32_t px = 0;
                if (lane_id() == 0) px = atomicAdd(&misc[CM_INSQ + par], 1u);
                px = __builtin_amdgcn_readfirstlane(px);
                if (px >= kCmpWaves) break;
                insert_unit(ring, tab, t0 + 512 * px + 8 * lane_id(), t1, n - kMfLimit, false, px == w && probe_h_valid, probe_h);
            }
            RSTAMP(4);
            // -- A4: tile k-1's bytes leave, 64 chunks a time, once all its strips are written out
            if (LZ4AMD_CMP_FLUSH_IN_A && prev_nstrips && !misc[CM_FAIL]) {
                while (uload_cm(&misc[CM_EMITDONE + par]) != (1u << prev_nstrips) - 1u) CMP_WAIT_PAUSE();
                RSTAMP(5);
                for (;;) {
                    uint32_t fx = 0;
                    if (lane_id() == 0) fx = atomicAdd(&misc[CM_FLUSHQ + par], 1u);
                    fx = __builtin_amdgcn_readfirstlane(fx);
                    if (fx >= kStageBytes / 1024) break;
                    const FlushCtx fj = flush_begin(smem, par ^ 1, a0, 64 * fx + lane_id());
                    if (64 * fx > fj.nfull) break;                      // (nothing of the tile that far: the chunks are clean)
                    flush_end(smem, fj, dst, a0, 64 * fx + lane_id());
                }
            }
            RSTAMP(6);
        }
        if (prof) { const uint64_t t = clock_ticks(); tp[4] += t - tq; tq = t; }
        // A full tile is done here - its pieces are in the table, the tile before has left, the next one's granules are in the ring - and has
        // no second barrier: the one at the next tile's top is the only one.  A small tile:
        const bool one_barrier = LZ4AMD_CMP_ONE_BARRIER && LZ4AMD_CMP_EARLY_COMMIT && LZ4AMD_CMP_FLUSH_IN_A && paired;
        bool do_flush = false;
        FlushCtx fc;
        if (!one_barrier) {
        __syncthreads();
        if (prof) { const uint64_t t = clock_ticks(); tp[2] += t - tq; tq = t; }
        // -- B: tile k-1's bytes leave; everybody inserts tile k into the table (positions that may start a match) - unless it was a full
        //    tile, which its writing waves inserted while the measuring ones were still at it
        do_flush = prev_nstrips && !misc[CM_FAIL] && !(LZ4AMD_CMP_FLUSH_IN_A && paired);      // (a full tile's waves stored the tile before already)
        if (do_flush) fc = flush_begin(smem, par ^ 1, a0, tid);
        if (!paired && n >= kMfLimit + 1) insert_unit(ring, tab, t0 + 8 * tid, t1, n - kMfLimit, small, probe_h_valid, probe_h);      // (t0 is a multiple of 16; tiles are at most 8 * kCmpThreads long; a full tile was inserted by its writing waves)
        }
#ifdef LZ4AMD_PROF_TILE
        if (prof) { const uint64_t t = clock_ticks(); if (tid == 0) ((uint64_t*)(smem + kCOffMisc))[14] += t - tq; }
#endif
        // the prefetched granules go into ring slots that hold bytes more than a window + a tile old (before the flush's store:
        // the wait for the load would wait for the store's acknowledgement as well - the counter is in order)
        if (!(LZ4AMD_CMP_EARLY_COMMIT && paired)) { if (Pp < pf_hi) ring_commit16(ring, Pp, pf); }
        if (pf_hi > loaded) loaded = (pf_hi + 15) & ~15u;
        if (do_flush) flush_end(smem, fc, dst, a0, tid);
        if (prof) { const uint64_t t = clock_ticks(); tp[3] += t - tq; tq = t; }
        prev_t0 = t0; prev_g0 = g0; prev_t1 = t1; prev_strip_len = paired ? 1024u : strip_len; prev_nstrips = paired ? 8u : nstrips; prev_rps = paired ? kRecsPerPair : kRecsPerStrip; par ^= 1;
        if (nstrips) tiles_parsed++;
        t0 = t1; tile_len = nt_len; strip_len = nt_strip;
    }
    __syncthreads();
    // -- the last tile's sequences (settled by wave 0 first)
    if (prev_nstrips) {
        if (w == kSettleWave) settle_tile(smem, par ^ 1, prev_nstrips, prev_g0, prev_t0, prev_strip_len, prev_t1, n, cap, a0);
        __syncthreads();
        const uint32_t ring_lo = loaded > kSrcRing ? loaded - kSrcRing : 0;
        if (w < prev_nstrips) emit_tile_strip(smem, par ^ 1, w, w, prev_rps, src, dst, a0, ring_lo, H);
        __syncthreads();
        if (!misc[CM_FAIL]) { const FlushCtx fc = flush_begin(smem, par ^ 1, a0, tid); flush_end(smem, fc, dst, a0, tid); }
    }
    __syncthreads();
    if (prof) {
        // developer aid: match + emit time of every wave (spread between the strips of a tile)
        if (w == 0 && lane_id() == 0) misc[CM_EMITQ] = (uint32_t)(tp[4] >> 4);
#ifndef LZ4AMD_PROF_TILE
        if (lane_id() == 0) misc[16 + w] = (uint32_t)((tp[1] + tp[4]) >> 4);
#endif
        __syncthreads();
        if (tid == 0) {
            uint64_t mx = 0, mn = ~0ull, sm = 0;
            for (uint32_t i = 0; i < kCmpWaves; i++) { const uint64_t v = (uint64_t)misc[16 + i] << 4; mx = v > mx ? v : mx; mn = v < mn ? v : mn; sm += v; }
#ifdef LZ4AMD_PROF_TILE
            { const uint64_t* m64 = (const uint64_t*)(smem + kCOffMisc); mx = m64[12]; mn = m64[13]; sm = m64[14] * kCmpWaves; }
#endif
#ifndef LZ4AMD_PROF_ROLES
            prof[0] = tp[0]; prof[1] = tp[1]; prof[2] = tp[2]; prof[3] = tp[3]; prof[4] = (uint64_t)misc[CM_EMITQ] << 4; prof[5] = mx; prof[6] = mn; prof[7] = sm / kCmpWaves;
#endif
#ifdef LZ4AMD_PROF_WAVES
            for (uint32_t i = 0; i < 8; i++) prof[i] = (uint64_t)misc[16 + 2 * i] | ((uint64_t)misc[17 + 2 * i] << 32);      // developer build: match + emit time of each wave (>> 4)
#endif
        }
    }
#ifdef LZ4AMD_PROF_ROLES
    if (prof && w == LZ4AMD_PROF_ROLES && lane_id() == 0) for (uint32_t i = 0; i < 8; i++) prof[i] = rt[i];
#endif
    // -- the pending bytes of the last chunk, then the final literal run (lz4.c:1302-1329)
    const uint32_t out = misc[CM_OUT], run = misc[CM_CARRY];
    const uint64_t total = (uint64_t)out + 1 + lit_hdr_ext(run) + run;
    if (misc[CM_FAIL] || total > cap) { if (tid == 0) { P.result[b] = 0; if (hints) *(uint32_t*)hints = 0; } return; }
    if (hints && tid == 0) {
        // the block's last sequence (its final literals) is a sequence like the others; the row behind the last one is the
        // block's end; row 0 (the block's first sequence) carries the number of rows; the header makes the table valid
        const uint32_t seqs = misc[CM_SEQS], nseq = seqs + 1, last_row = misc[CM_ROWS], nrows = last_row + 1;      // (the last sequence has a row of its own)
        if (last_row && last_row < H.cap_rows) st_hint(hints, last_row, out, n - run - pre, seqs);
        if (nrows <= H.cap_rows && !misc[CM_HOVER] && total < LZ4AMD_HINT_MAX_CSIZE) {
            hint_store_head(hints, (uint32_t)n_i, (uint32_t)total, nseq, nrows);
        } else *(uint32_t*)hints = 0;                                 // (more sequences than the table has room for: no table)
    }
    {
        const uint32_t r = (out + a0) & 15u, cfrom = misc[CM_TILE + 4 * par + T_CFROM];
        const uint8_t* Cn = (const uint8_t*)(smem + kCOffCarry) + 16 * par;
        if (tid >= cfrom && tid < r) dst[out - r + tid] = Cn[tid];
    }
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
