// lz4_hc_kernel.h -- batched LZ4 high-compression blocks (hash-chain match finder) for gfx950.
//
// Replaces, for a whole table of independent blocks resident in HBM, what the reference does per
// block in LZ4_compress_HC (lib/lz4hc.c:1519 -> LZ4HC_compress_generic 1388 -> LZ4HC_compress_hashChain
// 1121-1362) at the hash-chain levels 3..9: a 15-bit multiplicative hash of 4 bytes (lz4hc.c:121
// LZ4HC_hashPtr), a chain of 16-bit deltas to the previous position with the same hash
// (lz4hc.c:781-802 LZ4HC_Insert), up to nbSearches candidates per position inside the 64 KB window
// (k_clTable lz4hc.c:92-106: 4 << (level-3), 256 at level 9), lazy match selection.  The output is a
// legal LZ4 block (doc/lz4_Block_format.md); its bytes differ from the CPU library's, its size stays
// within about 2 % of the reference's level-9 result on datagen, text and binaries
// (tools/exp/hc_sim.c is the CPU model the scheme was sized with).
//
// Not a port.  The reference interleaves insert / search / lazy decisions in one serial loop; here the
// three are separate, each parallel in its own way, and one 1024-thread workgroup owns a block:
//
//   1 chain    all positions are linked first.  A wave takes 64 consecutive positions, resolves the
//              links inside the group with ballots, then - in position order, a token passed from
//              wave to wave through LDS - reads / updates the 32 K-entry head table (128 KB of LDS).
//              The chain (u16 delta per position) goes to a scratch array in HBM / L2.
//   2 search   EVERY position looks for its longest match (the parse order is not known yet, and
//              no longer matters).  The LZ4 window is 64 K positions x (1 source byte + 2 chain bytes)
//              = 192 KB, more than a CU's LDS, so the window is searched in BANDS of 32 K positions:
//              a pass streams the block through a 32 KB source ring + 64 KB chain ring in tiles of
//              8 K positions, every lane walks the chains of its 8 positions while the candidates are
//              inside the band; a walk whose next candidate lies beyond the band is parked as a list
//              entry {position, best length, best offset, distance of the next candidate, attempts
//              left} (8 bytes; a third of the positions on datagen -P60), and the next pass streams the
//              band 32 K further back and resumes the listed walks only.  In the nearest band a
//              candidate is compared over sixteen bytes at once (while the best match is shorter than
//              that the compare is the measurement); every probe of the inner loop is an LDS access.
//              Lengths are measured up to kHcLenCap bytes.
//   3 parse    16 strips, one wave each (the per-position results are staged through LDS in chunks
//              of 1 K positions): 64 positions at a time, every lane decides whether a sequence may
//              start at its position - a match that neither of the next two positions beats (lazy
//              evaluation, the role of the reference's LZ4HC_InsertAndGetWiderMatch retries,
//              lz4hc.c:1168-1330) - the wave walks from match end to next start over that bit mask,
//              extends capped matches with a wave-wide compare, and the taken lanes record {literal
//              run, offset, length}.  Matches end at the strip's end; pending literals carry over.
//   4 emit     strip sizes -> output offsets (wave 0), then every wave writes its strip
//              (emit_strip of the fast compressor).
// Levels 10-12 (lz4hc.c:92-106, LZ4HC_compress_optimal 1823-2130) replace phase 3's lazy choice by an optimal parse over the
// same per-position search results (hc_parse_strip_opt), with 96 / 512 / 2048 candidates per position in phase 2.
// Levels 1-2 (LZ4MID, lz4hc.c:472-773) replace phases 1 and 2 by two head tables and a search without chains or bands
// (hc_build_links_mid, hc_search_mid).
//
// HBM/L2 traffic per block: source read 2 + 2 x bands times, 2 B/position of chain written once and
// read once per band, 4 B/position of results written once (patched where a farther band finds better) and read once,
// 8 B per parked walk written and read per band, 8 B per sequence of records, compressed stream written once.
// Every position searches its own chain only: on copies-of-copies data, where a position's chain holds more true candidates than its
// attempts cover, the reference finds the far ones through the chains of later positions of the match and extends them backwards
// (lz4hc.c:885-960, 1168-1330), and its blocks are smaller (datagen -P99: by up to 20 % at a stream's start).
// No MFMA: integer / byte work.
#pragma once
#include "lz4_compress_kernel.h"

#ifndef LZ4AMD_HC_PROF
#if defined(LZ4AMD_PROF_ROLES) || defined(LZ4AMD_PROF_TILE) || defined(LZ4AMD_PROF_WAVES) || defined(LZ4AMD_PROF_HC)
#define LZ4AMD_HC_PROF 1
#else
#define LZ4AMD_HC_PROF 0        // 1: the phase stamps of tools/prof_cmp.py / prof_hc.py (LZ4AMD_PROF=1 in the environment): a developer build, tools/build_variant.sh prof -DLZ4AMD_HC_PROF=1
#endif
#endif
namespace lz4amd {

// developer counters of the CPU interpreter's build (tools/exp/hc_emu_stats.py): how many trips the walk loop makes, with how many lanes
#ifdef LZ4AMD_EMU_STATS
extern "C" unsigned long long lz4amd_emu_stats[16];
#define HC_STAT(i, v) __atomic_fetch_add(&lz4amd_emu_stats[i], (unsigned long long)(v), __ATOMIC_RELAXED)
#else
#define HC_STAT(i, v) ((void)0)
#endif

using HcBatch = ::lz4amd_hc_params;     // argument block (lz4amd_params.h)

enum : uint32_t {
    kHcThreads = 1024,
    kHcWaves = kHcThreads / 64,
    kHcHashLog = 15,                    // lz4hc.h:226 LZ4HC_HASH_LOG
#ifndef LZ4AMD_HC_TURN
#define LZ4AMD_HC_TURN 2
#endif
    kHcTurnGroups = LZ4AMD_HC_TURN,                  // groups of 64 positions a wave links per turn of the token
    kHcRing = 32768,                    // positions per band (source ring bytes, chain ring entries)
    kHcPad = 32,                        // mirror of the source ring's first bytes (reads of up to 36 bytes: see lds_ld16)
    kHcTile = 8192,                     // positions searched between two ring refills
    kHcPosPerThread = kHcTile / kHcThreads,
    kHcAhead = 272,                     // source bytes the rings hold past every position of the tile
    kHcLenCap = 250,                    // longest match a lane measures on its own (<= kHcAhead - 8, fits 8 bits)
    kHcBandStep = kHcRing - kHcAhead,   // how much further back the next band starts
    kHcBands = 3,                       // 3 x 32496 >= 65535 + kHcTile + kHcAhead: the whole LZ4 window, for every position
    kHcMinStrip = 1024,
    kHcEntCap = 6144,                   // parked walks of a tile a farther band stages at a time
    kHcChunk = 1024,                    // positions of search results a parsing wave stages in LDS at a time
#ifndef LZ4AMD_HC_BATCH
#define LZ4AMD_HC_BATCH 6
#endif
#ifndef LZ4AMD_HC_REFILL
#define LZ4AMD_HC_REFILL 16
#endif
#ifndef LZ4AMD_HC_RUN
#define LZ4AMD_HC_RUN 8
#endif
    kHcBatch = LZ4AMD_HC_BATCH,                       // links a lane chases before it verifies the candidates found (nearest band)
#ifndef LZ4AMD_HC_STALE
#define LZ4AMD_HC_STALE 32       // a walk of the nearest band ends after this many links in a row that did not lengthen its match (levels 3-9; 0: never).  Measured,
                                 // 1024 x 256 KiB at level 9, time / bytes against 0: 16 -10 % / 0 (-P60), -18 % / +0.27 % (-P90), -31 % / +1.6 % (-P99); 24 -8 / -13 / -26 %, 0 / +0.13 / +0.9 %;
                                 // 32 -6 / -7 / -19 %, 0 / +0.05 / +0.4 %; 64 -3 / -1 / -9 %; 128 0 / 0 / -3 %
#endif
#ifndef LZ4AMD_HC_BATCH_FAR
#define LZ4AMD_HC_BATCH_FAR 4
#endif
    kHcBatchFar = LZ4AMD_HC_BATCH_FAR,                //   ... in the farther bands
    kHcRun = LZ4AMD_HC_RUN,                         // consecutive positions a lane takes at a time (each inherits its predecessor's match)
    // An inherited match this long is kept without searching (the reference does not search inside a match it has taken either).  The
    // lazy parse loses next to nothing from 8 on (datagen -P60 / -P90 / -P20 blocks at level 9 against 32: +0.01 / +0.4 / +0.0 % bytes
    // for -14 % walk trips in the nearest band, tools/exp/hc_emu_stats.py); the optimal parse prices every cell and keeps 32.
    kHcSkipLenLazy = 8, kHcSkipLenOpt = 32,
    kHcRunsPerTile = kHcTile / kHcRun,
#ifndef LZ4AMD_HC_TAIL_POS
#define LZ4AMD_HC_TAIL_POS 4096
#endif
#ifndef LZ4AMD_HC_TAIL_RUN
#define LZ4AMD_HC_TAIL_RUN 2
#endif
    kHcTailPos = LZ4AMD_HC_TAIL_POS, kHcTailRun = LZ4AMD_HC_TAIL_RUN,      // the tile's last positions go out in short runs
    kHcLongRuns = (kHcTile - kHcTailPos) / kHcRun,
    kHcUnitsPerTile = kHcLongRuns + kHcTailPos / kHcTailRun,
};
// LDS carve-up (bytes); the chain phase, the search phase and the parse phase reuse the same region
enum : uint32_t {
    kHOffMisc = 0,                                  // u32[32]
    kHOffStrip = 128,                               // u32[8][16] strip summaries (rows S_N .. S_CARRY of the fast compressor's, row 6: sequences of the block before the strip)
    kHOffBody = 640,
    kHOffHead = kHOffBody,                          // phase 1: u32[1 << 15]
    kHOffSrc = kHOffBody,                           // phase 2: source ring + pad
    kHOffChain = kHOffSrc + kHcRing + kHcPad,       //          u16[kHcRing]
    kHOffMine = kHOffChain + 2 * kHcRing,           //          the tile's own bytes
    kHcMineBytes = kHcTile + kHcAhead + 48,           // 36-byte reads at up to kHcLenCap bytes past the tile's last position
    kHOffRes0 = kHOffMine + kHcMineBytes,           //          nearest band: u32[kHcTile] results of the tile; farther bands: HcEnt[kHcEntCap] parked walks
    kHcSearchEnd = kHOffRes0 + 8 * kHcEntCap,
    kHOffWtab = kHOffBody + (4u << kHcHashLog),     // phase 1: u32[kHcWaves][256] duplicate detection, one table per wave
    kHOffParse = kHOffBody,                         // phase 3: u32[kHcWaves][2 * kHcChunk]
    kHcLdsBytes = kHcSearchEnd,
};
static_assert(kHOffWtab + kHcWaves * 256 * 4 <= kHcLdsBytes, "head table + duplicate tables must fit");
static_assert(kHOffParse + kHcWaves * 2 * kHcChunk * 4 <= kHcLdsBytes, "parse staging must fit");
static_assert(kHcLdsBytes <= 160 * 1024, "one CU's LDS");
static_assert(kHcBands * kHcBandStep >= 65535 + kHcTile + kHcAhead, "bands must cover the LZ4 window");
static_assert(kHOffMine % 16 == 0 && kHOffRes0 % 16 == 0 && kHOffChain % 16 == 0, "16-byte LDS accesses");
static_assert(8 * kHcEntCap >= 4 * kHcTile, "the results of a tile and the staged entries share one region");
enum : uint32_t { HM_BLOCK = 0, HM_TOKEN = 1, HM_OUT = 2, HM_CARRY = 3, HM_FAIL = 4, HM_POOL = 5, HM_NLIST = 6, HM_HOVER = 7 /* a row of the entry-point table did not fit its room */, HM_SEQS = 8, HM_HK = 9 /* log2 of the table's row distance */, HM_FIRST0 = 16 /* [16]: first record of each strip's record area */ };

// scratch layout of one workgroup, for blocks of at most n bytes
__host__ __device__ inline uint64_t hc_chain_bytes(uint32_t n) { return ((uint64_t)2 * (n + 64) + 255) & ~255ull; }
__host__ __device__ inline uint64_t hc_st0_bytes(uint32_t n) { return ((uint64_t)4 * (n + kHcTile) + 255) & ~255ull; }
__host__ __device__ inline uint64_t hc_st1_bytes(uint32_t n) { return ((uint64_t)2 * (n + kHcTile) + 255) & ~255ull; }
__host__ __device__ inline uint64_t hc_recs_bytes(uint32_t n) { return (uint64_t)8 * (n / 4 + 64 * kHcWaves); }
__host__ __device__ inline uint64_t hc_list_bytes(uint32_t n) { return ((uint64_t)8 * (n + kHcTile) + 255) & ~255ull; }      // parked walks: at most one per position
__host__ __device__ inline uint64_t hc_count_bytes(uint32_t n) { return ((uint64_t)4 * (n / kHcTile + 2) + 255) & ~255ull; }
__host__ __device__ inline uint64_t hc_scratch_bytes(uint32_t n) { return hc_chain_bytes(n) + hc_st0_bytes(n) + hc_st1_bytes(n) + hc_recs_bytes(n) + hc_list_bytes(n) + hc_count_bytes(n); }

__device__ __forceinline__ uint32_t hc_attempts(int level) {
    // k_clTable lz4hc.c:92-106: levels 3..9 = 4..256 candidates per position; 10 / 11 = 96 / 512 (lz4hc.c:103-104); 12 = 2048 here
    // (lz4hc.c:105: 16384, behind a pattern analysis that keeps runs of a repeated pattern from costing that much - not built: a
    // position of such a run would hold its whole tile for 16384 dependent links).
    // (Levels 1 and 2 do not come here: hc_search_mid.)
    if (level < 1) level = 9;            // LZ4HC_CLEVEL_DEFAULT (lz4hc.c:110-113)
    if (level < 3) level = 3;
    if (level == 10) return 96u;
    if (level == 11) return 512u;
    if (level > 11) return 2048u;
    return 4u << (level - 3);
}
// A walk that is parked between two bands keeps its attempts in 8 bits: in units of 1 up to 256, of 2 up to 512, of 8 beyond
// (what is lost to the rounding is at most one unit per band).
__device__ __forceinline__ uint32_t hc_att_shift(uint32_t attempts) { return attempts <= 256u ? 0u : attempts <= 512u ? 1u : 3u; }
#ifndef LZ4AMD_HC_HASH_MUL
#define LZ4AMD_HC_HASH_MUL 2654435761u      // lz4hc.c:121 LZ4HC_hashPtr
#endif
__device__ __forceinline__ uint32_t hc_hash(uint32_t v) { return (v * LZ4AMD_HC_HASH_MUL) >> (32 - kHcHashLog); }

// Per-position search result: st0 = best length | offset << 8.  A walk that goes on in the next band is a list entry (hc_search_band).
struct HcEnt { uint32_t x, y; };

// ------------------------------------------------------------------------------ phase 1: chains
// Lanes of the group that share my hash: `below` = how many lower lanes do (-> the previous position with my
// hash is in the group), `last` = no higher lane does.  Most groups have no duplicates at all, so lanes first
// count themselves into a 1024-slot table private to the wave; only lanes whose slot was hit more than once
// compare hashes, one distinct value per trip.
__device__ __forceinline__ void hc_group_links(uint32_t h, bool valid, uint32_t* wtab, uint32_t lane, int& pred, bool& last) {
    const uint32_t word = (h >> 2) & 255u, sh = (h & 3u) * 8;      // 1024 one-byte counters in 256 words
    if (valid) atomicAdd(&wtab[word], 1u << sh);
    wave_lds_fence();
    const uint32_t cnt = valid ? (wtab[word] >> sh) & 0xFFu : 0u;
    wave_lds_fence();
    if (valid) wtab[word] = 0;
    wave_lds_fence();
    pred = -1; last = true;
    const bool crowded = cnt > 1;
    unsigned long long rem = __ballot(crowded);
    while (rem) {
        const uint32_t l = (uint32_t)__ffsll((long long)rem) - 1;
        const uint32_t hk = wave_readlane(h, l);
        const bool same = crowded && h == hk;
        const unsigned long long mm = __ballot(same);
        rem &= ~mm;
        if (same) {
            const unsigned long long below = mm & ((1ull << lane) - 1);
            pred = below ? 63 - __clzll(below) : -1;
            last = ((mm >> lane) >> 1) == 0;
        }
    }
}

__device__ __forceinline__ void hc_build_chain(lz4amd_gsrc src, uint32_t n, uint16_t* chain_g, char* smem) {
    const uint32_t tid = opaque_u32(threadIdx.x), lane = lane_here(), w = wave_id();
    uint32_t* misc = (uint32_t*)(smem + kHOffMisc);
    uint32_t* head = (uint32_t*)(smem + kHOffHead);
    uint32_t* wtab = (uint32_t*)(smem + kHOffWtab) + w * 256;
    for (uint32_t i = tid; i < (1u << kHcHashLog); i += kHcThreads) head[i] = 0;
    for (uint32_t i = lane; i < 256; i += 64) wtab[i] = 0;
    if (tid == 0) misc[HM_TOKEN] = 0;
    __syncthreads();
    const uint32_t ngroups = (n + 63) / 64;
    const uint32_t nturns = (ngroups + kHcTurnGroups - 1) / kHcTurnGroups;
    for (uint32_t turn = w; turn < nturns; turn += kHcWaves) {
        uint32_t h[kHcTurnGroups], prev1[kHcTurnGroups];
        bool valid[kHcTurnGroups], first[kHcTurnGroups], last[kHcTurnGroups];
        // links inside each group of 64: previous lane with my hash, and whether I am the last one with it
#pragma unroll
        for (uint32_t j = 0; j < kHcTurnGroups; j++) {
            const uint32_t p = (turn * kHcTurnGroups + j) * 64 + lane;
            valid[j] = p + 4 <= n;
            uint32_t v = 0;
            if (valid[j]) __builtin_memcpy(&v, src + p, 4);
            h[j] = hc_hash(v);
            int pred;
            hc_group_links(h[j], valid[j], wtab, lane, pred, last[j]);
            first[j] = pred < 0;
            prev1[j] = pred >= 0 ? (turn * kHcTurnGroups + j) * 64 + (uint32_t)pred + 1 : 0;
        }
        // my turn: groups read and update the head table in position order
        while (lds_load_acquire(&misc[HM_TOKEN]) != turn) spin_pause();
#pragma unroll
        for (uint32_t j = 0; j < kHcTurnGroups; j++) {
            const uint32_t p = (turn * kHcTurnGroups + j) * 64 + lane;
            if (valid[j] && first[j]) prev1[j] = head[h[j]];
            wave_lds_order();
            if (valid[j] && last[j]) head[h[j]] = p + 1;
            wave_lds_order();
        }
        // reads, writes and the token are queued back to back: the DS unit keeps a wave's order, the next
        // wave's accesses come after the token it has seen
        if (lane == 0) lds_store_relaxed(&misc[HM_TOKEN], turn + 1);
#pragma unroll
        for (uint32_t j = 0; j < kHcTurnGroups; j++) {
            const uint32_t p = (turn * kHcTurnGroups + j) * 64 + lane;        // prev1 = previous position + 1, 0 = none
            uint32_t delta = 0;
            if (valid[j] && prev1[j]) { const uint32_t d = p + 1 - prev1[j]; if (d <= kMaxDistance) delta = d; }
            if (p < ngroups * 64) chain_g[p] = (uint16_t)delta; // the array is padded to a multiple of 64 entries
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------ phase 2: band search
// sixteen bytes at any alignment out of an LDS byte array: five aligned dwords + four v_alignbyte
struct Q16 { uint32_t a, b, c, d; };
__device__ __forceinline__ Q16 lds_ld16(const uint8_t* base, uint32_t o) {
    const uint32_t* w = (const uint32_t*)(base + (o & ~3u));
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4], sh = o & 3u;
    Q16 r; r.a = align_bytes(w1, w0, sh); r.b = align_bytes(w2, w1, sh); r.c = align_bytes(w3, w2, sh); r.d = align_bytes(w4, w3, sh);
    return r;
}
// number of equal leading bytes of two 16-byte strings (0..16).  No branches: the first set bit of every dword's difference (-1, i.e.
// 0xFFFFFFFF, where there is none: __ffs(0) - 1), as a byte index, and the minimum of the four.
__device__ __forceinline__ uint32_t equal_bytes16(const Q16& x, const Q16& y) {
    const uint32_t f0 = (uint32_t)(__ffs((int)(x.a ^ y.a)) - 1) >> 3, f1 = ((uint32_t)(__ffs((int)(x.b ^ y.b)) - 1) >> 3) + 4;
    const uint32_t f2 = ((uint32_t)(__ffs((int)(x.c ^ y.c)) - 1) >> 3) + 8, f3 = ((uint32_t)(__ffs((int)(x.d ^ y.d)) - 1) >> 3) + 12;
    const uint32_t m01 = f0 < f1 ? f0 : f1, m23 = f2 < f3 ? f2 : f3, m = m01 < m23 ? m01 : m23;
    return m < 16 ? m : 16;
}
// four bytes at any alignment out of an LDS byte array (two aligned dword reads + v_alignbyte)
__device__ __forceinline__ uint32_t lds_ld4(const uint8_t* base, uint32_t o) {
    const uint32_t* a = (const uint32_t*)(base + (o & ~3u));
    return align_bytes(a[1], a[0], o & 3u);
}
__device__ __forceinline__ void hc_commit_src(uint8_t* ring, uint32_t P, const U32x4& v) {
    const uint32_t o = P & (kHcRing - 1);
    *(U32x4*)(ring + o) = v;
    if (o < kHcPad) *(U32x4*)(ring + kHcRing + o) = v;
}

// common length of the strings at ring offset qo and at `mine` offset pp, given that the first l bytes are
// equal; 32 bytes per trip (all reads of a trip are independent), at most lim
__device__ __forceinline__ uint32_t hc_count(const uint8_t* ring, const uint8_t* mine, uint32_t qo, uint32_t pp, uint32_t l, uint32_t lim) {
    if (l < lim) {                                              // most matches end inside the first 16 bytes
        const uint32_t e = equal_bytes16(lds_ld16(ring, (qo + l) & (kHcRing - 1)), lds_ld16(mine, pp + l));
        l += e;
        if (e == 16) {
            while (l < lim) {
                const Q16 r0 = lds_ld16(ring, (qo + l) & (kHcRing - 1)), r1 = lds_ld16(ring, (qo + l + 16) & (kHcRing - 1));
                const Q16 m0 = lds_ld16(mine, pp + l), m1 = lds_ld16(mine, pp + l + 16);
                const uint32_t e0 = equal_bytes16(r0, m0);
                if (e0 < 16) { l += e0; break; }
                const uint32_t e1 = equal_bytes16(r1, m1);
                l += 16 + e1;
                if (e1 < 16) break;
            }
        }
    }
    return l > lim ? lim : l;
}

// What a walk leaves behind between two bands is a list entry, not a slot of a per-position array: most walks end in the nearest
// band, so the farther bands read a third of the positions (and the last one a sixth) instead of streaming 6 bytes of state per
// position in and out.  Entry (8 bytes): e0 = position in the tile | distance of the next candidate << 16, e1 = best offset |
// best length << 16 | attempts left (hc_att_shift units) << 24.  A tile's entries sit at list_g[t0 ...], their number in
// count_g[tile]; a band rewrites a tile's list in place (it has staged what it overwrites).  The per-position results (best
// length | offset << 8) are written densely by the nearest band and patched by the farther ones where a walk found a longer match.
template <bool NEAR> __device__ __forceinline__ void hc_search_band(lz4amd_gsrc src, uint32_t n, uint32_t first, const uint16_t* chain_g, uint32_t* st0_g, HcEnt* list_g, uint32_t* count_g,
                                               uint32_t band, uint32_t attempts, uint32_t skip_len, bool favor, char* smem, uint64_t* prof = nullptr) {
    const uint32_t tid = opaque_u32(threadIdx.x);
    constexpr uint32_t kB = NEAR ? (uint32_t)kHcBatch : (uint32_t)kHcBatchFar;      // links a lane chases per trip
#ifdef LZ4AMD_PROF_HC
    uint64_t hp_trips = 0, hp_lanes = 0, hp_loop = 0, hp_wait = 0, hp_hits = 0, hp_hit_lanes = 0, hp_t0x = 0;
#else
    (void)prof;
#endif
    uint8_t* ring = (uint8_t*)(smem + kHOffSrc);
    uint16_t* cring = (uint16_t*)(smem + kHOffChain);
    uint8_t* mine = (uint8_t*)(smem + kHOffMine);
    uint32_t* res0 = (uint32_t*)(smem + kHOffRes0);                     // nearest band: the tile's results
    HcEnt* ent = (HcEnt*)(smem + kHOffRes0);                            // farther bands: the staged entries
    uint32_t* misc = (uint32_t*)(smem + kHOffMisc);
    const uint32_t n64 = (n + 63) & ~63u;
    const int32_t last_q = (int32_t)n - (int32_t)kMfLimit;            // last position that may start a match
    const int32_t shift = (int32_t)(band * kHcBandStep);
    const bool final_band = band + 1 == kHcBands;
    const uint32_t ash = hc_att_shift(attempts);
    const uint32_t dmin = favor ? 8u : 1u;                              // the smallest distance of a candidate that counts (cd[k] = 0: none)
    // the rings hold positions [H - kHcRing, H), H = t0 + kHcTile + kHcAhead - shift
    // -- first tile: fill [0, H(0)) directly
    {
        const int32_t H = (int32_t)(kHcTile + kHcAhead) - shift;
        const int32_t Ps = 16 * (int32_t)tid, Pc = 8 * (int32_t)tid;
        if (Ps < H && (uint32_t)Ps < n) hc_commit_src(ring, (uint32_t)Ps, load_src16(src, n, (uint32_t)Ps));
        for (int32_t P = Pc; P < H && (uint32_t)P < n64; P += 8 * (int32_t)kHcThreads)
            *(U32x4*)(cring + (P & (int32_t)(kHcRing - 1))) = *(const U32x4*)(chain_g + P);
        if (tid < kHcMineBytes / 16) *(U32x4*)(mine + 16 * tid) = load_src16(src, n, 16 * tid);      // zero filled past n
    }
    if (tid == 0) misc[HM_NLIST] = 0;
    for (uint32_t t0 = 0; t0 < n; t0 += kHcTile) {
        const int32_t H = (int32_t)(t0 + kHcTile + kHcAhead) - shift;
        const int32_t low = H - (int32_t)kHcRing;                     // lowest candidate position of this band
        // -- prefetch the next tile's ring granules and own bytes
        const int32_t Ps = H + 16 * (int32_t)tid, Pc = H + 8 * (int32_t)tid;
        const bool have_s = tid < kHcTile / 16 && Ps >= 0 && (uint32_t)Ps < n;
        const bool have_c = Pc >= 0 && (uint32_t)Pc < n64;           // kHcTile / 8 == kHcThreads granules
        const uint32_t Pm = t0 + kHcTile + 16 * tid;
        const bool have_m = tid < kHcMineBytes / 16;
        U32x4 ps, pc, pm;
        ps[0] = ps[1] = ps[2] = ps[3] = 0; pc = ps; pm = ps;
        if (have_s) ps = load_src16(src, n, (uint32_t)Ps);
        if (have_c) pc = *(const U32x4*)(chain_g + Pc);
        if (have_m && Pm < n) pm = load_src16(src, n, Pm);
        // -- farther bands: the tile's parked walks, staged kHcEntCap at a time (one round, unless nearly every position of the tile
        //    is still walking); the nearest band: one round over the tile's runs
        HcEnt* const list_t = list_g + t0;                           // the tile's entries
        const uint32_t cnt = NEAR ? 0u : count_g[t0 / kHcTile];
        uint32_t c0 = 0;
        do {
            uint32_t cn = 0;
            if (!NEAR) {
                cn = cnt - c0 < kHcEntCap ? cnt - c0 : kHcEntCap;
                for (uint32_t i = tid; i < cn; i += kHcThreads) ent[i] = list_t[c0 + i];
            }
            if (tid == 0) misc[HM_POOL] = 0;
            __syncthreads();
            // (the nearest band's last kHcTailPos positions of a tile are handed out in runs of kHcTailRun: what a tile waits for at its barrier is the run that
            //  was taken last, walked by one lane position after position)
            const uint32_t nunits = NEAR ? (uint32_t)kHcUnitsPerTile : cn;          // work units of the round: entries, or runs
            // -- the walks.  The nearest band's tile is a pool of runs of kHcRun consecutive positions; idle lanes of any wave take
            //    the next runs (which lane walks a run does not change its result).  The loop is wave-synchronous
            //    and predicated: every trip CHASES up to kB links of each lane's chain (dependent LDS reads,
            //    nothing else on the path), then VERIFIES the candidates found (independent reads).
            {
                const uint32_t lane = lane_here();
                bool pool_dry = nunits == 0;                                 // the round's pool is empty
                uint32_t run_left = 0;                                       // positions of my run still to start
                uint32_t eidx = 0;                                           // (farther bands) my entry
                uint32_t inh_len = 0, inh_off = 0;                           // what the previous position of my run leaves to the next
                bool inh_capped = false;
                bool active = false;
                int32_t p = 0;
                uint32_t dist = 0, best = 3, boff = 0, att = 0, lim = 0, mt = 0, pp = 0, best_in = 3;
                uint32_t stale = 0;                                          // (LZ4AMD_HC_STALE) links since the match last grew
                Q16 mw; mw.a = mw.b = mw.c = mw.d = 0;                        // sixteen of my own bytes: the window the candidates are compared in
#ifdef LZ4AMD_PROF_HC
                hp_t0x = clock_ticks();
#endif
                for (;;) {
                    // ---- hand out work to idle lanes, from one pool for the whole round
                    //      (an LDS counter; asked only when a quarter of the wave is idle)
                    const bool want = !active && run_left == 0;
                    const unsigned long long idle = __ballot(want);
                    const uint32_t nidle = (uint32_t)__popcll(idle);
                    if (!pool_dry && nidle >= LZ4AMD_HC_REFILL) {                   // (when nobody works all 64 lanes are idle: no test of its own)
                        uint32_t base = 0;
                        if (lane == (uint32_t)__ffsll((long long)idle) - 1) base = atomicAdd(&misc[HM_POOL], nidle);
                        base = wave_readlane(base, (uint32_t)__ffsll((long long)idle) - 1);
                        const uint32_t mine_i = base + lanes_below(idle);
                        if (base + nidle >= nunits) pool_dry = true;
                        if (want && mine_i < nunits) {
                            if (!NEAR) { eidx = mine_i; run_left = 1; }
                            else if (mine_i < kHcLongRuns) { pp = mine_i * kHcRun - 1; run_left = kHcRun; }
                            else { pp = kHcLongRuns * kHcRun + (mine_i - kHcLongRuns) * kHcTailRun - 1; run_left = kHcTailRun; }
                            inh_len = 0;
                        }
                    }
                    // (Round 5 let idle lanes take over the last queued position of busy ones once the pool was dry.  A position taken over starts without
                    //  what its predecessor found, so its result depended on which lanes happened to be idle - harmless while every walk was exhaustive,
                    //  visible in the bytes once walks end early (LZ4AMD_HC_STALE) - and with the tile's last positions in short runs the kernel is 3 %
                    //  faster without it.)
                    // ---- next position of my run.  In the nearest band it starts from what its predecessor found:
                    //      a match of length L at p is a match of length L - 1 at p + 1 (same offset), so inside a long
                    //      match only the first position pays for measuring it
                    if (!active && run_left) {
                        run_left--;
                        bool walk = true, kept = false;
                        if (NEAR) {
                            pp++;
                            p = (int32_t)(t0 + pp);
                            if (p > last_q || (uint32_t)p < first) { res0[pp] = 0; walk = false; }     // the history, and the slots past the block's end
                            else { dist = cring[(uint32_t)p & (kHcRing - 1)]; best = 3; boff = 0; att = attempts; }
                        } else {
                            const HcEnt e = ent[eidx];
                            pp = e.x & 0xFFFFu; p = (int32_t)(t0 + pp);
                            dist = e.x >> 16; boff = e.y & 0xFFFFu; best = (e.y >> 16) & 0xFFu; att = ((e.y >> 24) << ash) + 1;
                        }
                        if (walk) {
                            lim = (uint32_t)((int32_t)n - (int32_t)kLastLiterals - p);
                            if (lim > kHcLenCap) lim = kHcLenCap;
                            best_in = best;
                            if (NEAR && inh_len >= kMinMatch && inh_len > best) {     // (nearest band only: a farther band's work units are single positions)
                                best = inh_len; boff = inh_off;
                                // a capped predecessor may match further than it measured
                                if (inh_capped) best = hc_count(ring, mine, (uint32_t)(p - (int32_t)boff) & (kHcRing - 1), pp, best, lim);
                                // nothing can beat a full-length match; and inside a long match the positions after the
                                // first are not searched at all (the reference does not visit them either)
                                if (best >= lim || best >= skip_len) {
                                    res0[pp] = best | (boff << 8);
                                    inh_len = best - 1; inh_capped = best >= lim;
                                    walk = false; kept = true;
                                }
                            }
                        }
                        if (walk) { if (NEAR) mw = lds_ld16(mine, pp + (best > 15 ? best - 15 : 0)); else mt = lds_ld4(mine, pp + best - 3); active = true; stale = 0; HC_STAT(NEAR ? 0 : 5, 1); }
                        else if (!kept) inh_len = 0;                        // nothing to hand to the next position
                        if (kept) HC_STAT(NEAR ? 1 : 6, 1);
                    }
                    if (!__ballot(active)) { if (pool_dry && !__ballot(run_left != 0)) break; continue; }
#ifdef LZ4AMD_PROF_HC
                    hp_trips++; hp_lanes += (uint32_t)__popcll(__ballot(active));
#endif
                    if (lane == 0) HC_STAT(NEAR ? 2 : 7, 1);
                    if (active) HC_STAT(NEAR ? 3 : 8, 1);
                    // ---- chase: distances of the next candidates; `dist` = the one to look at next.  A candidate counts while it is inside
                    //      the window AND the band: one compare against the higher of the two bounds; which of them ended the walk is
                    //      looked at once, behind the links
                    uint32_t cd[kB];
                    const uint32_t att_trip0 = att, best_trip0 = best;
                    const int32_t qmin = p - (int32_t)kMaxDistance > low ? p - (int32_t)kMaxDistance : low;
                    bool alive = active && dist != 0 && att != 0;
#pragma unroll
                    for (uint32_t k = 0; k < kB; k++) {
                        const int32_t q = p - (int32_t)dist;
                        const bool take = alive && q >= qmin;
                        cd[k] = take ? dist : 0;
                        const uint32_t d = cring[(uint32_t)q & (kHcRing - 1)];
                        att -= take ? 1u : 0u;
                        alive = take && d != 0 && att != 0;
                        dist += take ? d : 0u;
                    }
                    // the walk is over unless its chain goes on inside the band; a candidate beyond the band (not beyond the window, and
                    // with attempts left) is where the next band resumes
                    bool over = !alive;
                    uint32_t next = 0;
                    if (!alive && !final_band && att != 0 && dist != 0 && dist <= kMaxDistance && p - (int32_t)dist < low) next = dist;
                    bool full = false;
                    if (NEAR) {
                        // ---- verify: sixteen bytes of every candidate against my own, in the window that ends at index `best` at the latest
                        //      (w0 = best - 15; 0 while best is below 16, and then the compare IS the measurement: lz4hc.c:934-946).  Only a
                        //      candidate that agrees on the whole window can be longer than that, and is measured out in the loop below.
                        const uint32_t w0 = best > 15 ? best - 15 : 0;
                        Q16 cw[kB];
#pragma unroll
                        for (uint32_t k = 0; k < kB; k++)
                            cw[k] = lds_ld16(ring, ((uint32_t)(p - (int32_t)cd[k]) + w0) & (kHcRing - 1));
                        uint32_t ext = 0;                                        // candidates that agree on all sixteen, as a bit mask per lane
#pragma unroll
                        for (uint32_t k = 0; k < kB; k++) {
                            const bool cand = cd[k] >= dmin;                        // (a candidate; favorDecSpeed skips offsets below 8, lz4hc.c:926-929)
                            const uint32_t e = cand ? equal_bytes16(cw[k], mw) : 0u;
                            ext |= e == 16 ? 1u << k : 0u;
                            const uint32_t ec = e < lim ? e : lim;
                            if (w0 == 0 && e < 16 && ec > best) { best = ec; boff = cd[k]; if (ec >= lim) full = true; }
                        }
                        if (full) ext = 0;
                        while (__ballot(ext != 0)) {
                            if (lane == 0) HC_STAT(NEAR ? 4 : 9, 1);
#ifdef LZ4AMD_PROF_HC
                            hp_hits++; hp_hit_lanes += (uint32_t)__popcll(__ballot(ext != 0));
#endif
                            if (ext) {
                                const uint32_t k = (uint32_t)__ffs((int)ext) - 1;
                                ext &= ext - 1;
                                uint32_t cdk = cd[0];
#pragma unroll
                                for (uint32_t j = 1; j < kB; j++) cdk = k == j ? cd[j] : cdk;
                                const uint32_t qo = (uint32_t)(p - (int32_t)cdk) & (kHcRing - 1);
                                // an earlier candidate of the batch may have raised `best` past the window: test again at the new index
                                if (best <= w0 + 15 || lds_ld4(ring, (qo + best - 3) & (kHcRing - 1)) == lds_ld4(mine, pp + best - 3)) {
                                    const uint32_t l = hc_count(ring, mine, qo, pp, w0 == 0 ? 16u : 0u, lim);
                                    if (l > best) {
                                        best = l; boff = cdk;
                                        if (l >= lim) { full = true; ext = 0; }
                                    }
                                }
                            }
                        }
                        {   // my own bytes in the window of the next trip
                            const uint32_t w1 = best > 15 ? best - 15 : 0;
                            if (w1 != w0) mw = lds_ld16(mine, pp + w1);
                        }
                    } else {
                        // farther bands: a walk that comes this far mostly has its match already, and nearly every candidate fails the
                        // cheapest test there is - the four bytes that end at index `best` (for best = 3 the MINMATCH test; lz4hc.c:934-936)
                        uint32_t ct[kB];
#pragma unroll
                        for (uint32_t k = 0; k < kB; k++)
                            ct[k] = lds_ld4(ring, ((uint32_t)(p - (int32_t)cd[k]) + best - 3) & (kHcRing - 1));
                        const uint32_t best0 = best;                            // what the ct[] were read against
                        // candidates that pass, as a bit mask per lane; the expensive part (measuring a match) is entered
                        // once per trip for every lane's nearest passing candidate, again only for lanes that have another
                        uint32_t hits = 0;
#pragma unroll
                        for (uint32_t k = 0; k < kB; k++) hits |= (cd[k] >= dmin && ct[k] == mt) ? 1u << k : 0u;      // (favorDecSpeed skips offsets below 8, lz4hc.c:926-929)
                        while (__ballot(hits != 0)) {
                            if (lane == 0) HC_STAT(NEAR ? 4 : 9, 1);
#ifdef LZ4AMD_PROF_HC
                            hp_hits++; hp_hit_lanes += (uint32_t)__popcll(__ballot(hits != 0));
#endif
                            if (hits) {
                                const uint32_t k = (uint32_t)__ffs((int)hits) - 1;
                                hits &= hits - 1;
                                uint32_t cdk = cd[0];
#pragma unroll
                                for (uint32_t j = 1; j < kB; j++) cdk = k == j ? cd[j] : cdk;
                                const uint32_t qo = (uint32_t)(p - (int32_t)cdk) & (kHcRing - 1);
                                // an earlier candidate of the batch may have raised `best`: test again at the new index
                                if (best == best0 || lds_ld4(ring, (qo + best - 3) & (kHcRing - 1)) == mt) {
                                    const uint32_t l = hc_count(ring, mine, qo, pp, 0, lim);
                                    if (l > best) {
                                        best = l; boff = cdk;
                                        if (l >= lim) { full = true; hits = 0; }
                                        else mt = lds_ld4(mine, pp + best - 3);
                                    }
                                }
                            }
                        }
                    }
                    if (full) { over = true; next = 0; }
#if LZ4AMD_HC_STALE
                    // A chain inside a stretch of copies of copies holds candidate after candidate with the same bytes: none of them lengthens the
                    // match, and a walk through all of them (256 links at level 9) is what a tile waits for at its barrier - one position in a
                    // thousand.  The nearest band gives such a walk up after LZ4AMD_HC_STALE links in a row without gain (a rule of the position
                    // alone: the output does not depend on which lane walks it).
                    if (NEAR && attempts <= 256) {                     // (the optimal parse of levels 10-12 asks for deeper searches on purpose)
                        stale = best > best_trip0 ? 0u : stale + (att_trip0 - att);
                        if (active && stale >= LZ4AMD_HC_STALE) { over = true; next = 0; }
                    }
#endif
                    const bool park = active && over && next != 0;           // the walk goes on in the next band
                    if (active && over) {
                        if (NEAR) res0[pp] = best | (boff << 8);
                        else if (best > best_in) st0_g[p] = best | (boff << 8);
                        // what the next position of my run may start from
                        inh_len = best > kMinMatch ? best - 1 : 0; inh_off = boff; inh_capped = best >= lim;
                        active = false;
                    }
                    const unsigned long long parked = __ballot(park);        // (one counter update per wave)
                    if (parked) {
                        const uint32_t leader = (uint32_t)__ffsll((long long)parked) - 1;
                        uint32_t base = 0;
                        if (lane == leader) base = atomicAdd(&misc[HM_NLIST], (uint32_t)__popcll(parked));
                        base = wave_readlane(base, leader);
                        if (park) { HcEnt e; e.x = pp | (next << 16); e.y = boff | (best << 16) | (((att - 1) >> ash) << 24); list_t[base + lanes_below(parked)] = e; }
                    }
                }
            }
#ifdef LZ4AMD_PROF_HC
            const uint64_t hp_t1 = clock_ticks();
            hp_loop += hp_t1 - hp_t0x;
#endif
            __syncthreads();
#ifdef LZ4AMD_PROF_HC
            { const uint64_t t2 = clock_ticks(); hp_wait += t2 - hp_t1; }
#endif
            c0 += kHcEntCap;
        } while (c0 < cnt);
        // -- the nearest band flushes the tile's results (coalesced); the prefetched granules are committed: they replace positions
        //    below the next tile's band
        if (NEAR) {
            *(U32x4*)(st0_g + t0 + 4 * tid) = *(const U32x4*)(res0 + 4 * tid);
            *(U32x4*)(st0_g + t0 + 4 * (tid + kHcThreads)) = *(const U32x4*)(res0 + 4 * (tid + kHcThreads));
        }
        if (tid == 0) { if (!final_band) count_g[t0 / kHcTile] = misc[HM_NLIST]; misc[HM_NLIST] = 0; }
        if (have_s) hc_commit_src(ring, (uint32_t)Ps, ps);
        if (have_c) *(U32x4*)(cring + ((uint32_t)Pc & (kHcRing - 1))) = pc;
        if (have_m) *(U32x4*)(mine + 16 * tid) = pm;
    }
    __syncthreads();
#ifdef LZ4AMD_PROF_HC
    if (prof && NEAR && tid == 0) { prof[3] += hp_trips | (hp_lanes << 32); prof[6] += hp_hits | (hp_hit_lanes << 32); prof[7] += (hp_loop >> 4) | ((hp_wait >> 4) << 32); }
#endif
}

// ------------------------------------------------------------------------------ phase 3: parse (one strip)
// 8 source bytes at position a (a < n), zero filled past the end of the block
__device__ __forceinline__ uint64_t hc_ld8(lz4amd_gsrc src, uint32_t n, uint32_t a) {
    if (a + 8 <= n) { uint64_t v; __builtin_memcpy(&v, src + a, 8); return v; }
    uint64_t v = 0;
#pragma nounroll
    for (uint32_t i = 0; i < 8 && a + i < n; i++) v |= (uint64_t)src[a + i] << (8 * i);
    return v;
}

// The lazy rule is local to a position - a match at f is taken unless one of the next two positions has a better one - so every
// lane decides for its own position at once (`good`), and what is left of the reference's serial loop is a walk over a bit mask:
// from the end of a taken match to the next good position.  The walk costs a handful of scalar instructions per sequence; the
// literal runs, the record writes and the encoded sizes of all sequences of the window are then worked out by their own lanes.
__device__ __forceinline__ void hc_parse_strip(lz4amd_gsrc src, uint32_t n, const uint32_t* best_g, MatchRec* recs,
                                               uint32_t* strip, uint32_t* stage, uint32_t w, uint32_t cs, uint32_t ce) {
    const uint32_t lane = lane_here();
    uint32_t nseq = 0, anchor = cs;
    uint32_t enc_l = 0, ll0_l = 0;                                      // per lane; summed over the wave at the end
    if (n >= kMfLimit + 1 && cs <= n - kMfLimit) {
        const uint32_t last_q = n - kMfLimit;
        uint32_t mlimit = n - kLastLiterals; if (mlimit > ce) mlimit = ce;
        uint32_t ip = cs;
        uint32_t staged = cs;                                           // results of [staged - 2 * kHcChunk, staged) are in `stage`
        while (ip < ce && ip <= last_q) {
            while (ip + 64 > staged) {                                  // stage the next chunk (cs is a multiple of 64; the array is padded)
                uint32_t* dstp = stage + ((staged - cs) & (2 * kHcChunk - 1));
#pragma unroll
                for (uint32_t k = 0; k < kHcChunk / 256; k++)
                    *(U32x4*)(dstp + 4 * (lane + 64 * k)) = *(const U32x4*)(best_g + staged + 4 * (lane + 64 * k));
                staged += kHcChunk;
                wave_lds_fence();
            }
            const uint32_t pos = ip + lane;
            uint32_t len = 0, off = 0;
            if (pos <= last_q && pos + kMinMatch <= mlimit) {
                const uint32_t s = stage[(pos - cs) & (2 * kHcChunk - 1)];
                const uint32_t l = s & 0xFFu;
                if (l >= kMinMatch) { len = l; off = s >> 8; }
            }
            // -- every position: may a sequence start here (the two positions after it must be in the window), and where would it end
            const uint32_t l1 = wave_next_u32(len), l2 = wave_next_u32(l1);
            const bool good = len != 0 && lane < 62 && !(l1 > len || l2 > len + 1);      // a later start is better: the position becomes a literal
            uint32_t ml = len;
            if (pos + ml > mlimit) ml = mlimit - pos;                   // (len != 0: at least kMinMatch are left)
            const bool capped = good && len >= kHcLenCap && pos + len < mlimit;      // the search stopped measuring: see below
            const uint32_t jend = lane + ml;
            const unsigned long long G = __ballot(good), C = __ballot(capped), M = __ballot(len != 0);
            // -- the walk: first good position at or after the end of the match before it
            unsigned long long T = 0;                                    // the positions whose match is taken
            uint32_t a = 0;                                              // (ip is never below the anchor)
            uint32_t capf = 64;
            for (;;) {
                const unsigned long long rem = G & (~0ull << a);
                if (!rem) break;
                const uint32_t f = (uint32_t)__ffsll((long long)rem) - 1;
                if ((C >> f) & 1) { capf = f; break; }
                T |= 1ull << f;
                a = wave_readlane(jend, f);
                if (a >= 64) break;
            }
            uint32_t next_ip;
            if (a >= 64) next_ip = ip + a;
            else {                                                       // positions 62 and 63 are decided by the next window
                const uint32_t nx = ((M >> 62) & 1) ? 62u : (M >> 63) ? 63u : 64u;
                next_ip = ip + (a > nx ? a : nx);
            }
            // -- the taken sequences, each by its own lane
            if (T) {
                const bool taken = (T >> lane) & 1;
                const uint32_t pm = wave_prev_u32(wave_incl_max_u32(taken ? pos + ml : 0u));      // where the taken match before mine ends
                if (taken) {
                    const uint32_t ll = pos - (pm > anchor ? pm : anchor);
                    const uint32_t idx = nseq + lanes_below(T);
                    MatchRec r; r.ll = ll; r.mo = off | ((ml - kMinMatch) << 16); recs[idx] = r;
                    enc_l += enc_size(ll, ml - kMinMatch);
                    if (idx == 0) ll0_l = ll;
                }
                nseq += (uint32_t)__popcll(T);
                anchor = ip + a;
            }
            if (capf < 64) {
                // capped by the search: wave-wide compare, 8 bytes per lane per trip (lz4.c:680-703 LZ4_count)
                const uint32_t x = ip + capf, of = wave_readlane(off, capf);
                uint32_t mlx = wave_readlane(len, capf);
                for (;;) {
                    const uint32_t p8 = x + mlx + 8 * lane;
                    uint32_t same = 0;
                    if (p8 < mlimit) {
                        same = equal_bytes8(hc_ld8(src, n, p8), hc_ld8(src, n, p8 - of));
                        if (same > mlimit - p8) same = mlimit - p8;
                    }
                    const unsigned long long brk = __ballot(same < 8);
                    if (brk) {
                        const uint32_t fl = (uint32_t)__ffsll((long long)brk) - 1;
                        mlx += 8 * fl + wave_readlane(same, fl);
                        break;
                    }
                    mlx += 512;
                }
                if (x + mlx > mlimit) mlx = mlimit - x;
                if (mlx > 65535u) mlx = 65535u;                           // the record keeps ml - 4 in 16 bits
                const uint32_t ll = x - anchor;
                if (lane == 0) {
                    MatchRec r; r.ll = ll; r.mo = of | ((mlx - kMinMatch) << 16); recs[nseq] = r;
                    enc_l += enc_size(ll, mlx - kMinMatch);
                    if (nseq == 0) ll0_l = ll;
                }
                nseq++;
                anchor = x + mlx;
                next_ip = anchor;                                        // (at least kHcLenCap past x)
            }
            // a jump past the staged chunks (a long match): restart the staging at the landing chunk
            if (next_ip >= staged + kHcChunk) staged = cs + ((next_ip - cs) & ~(kHcChunk - 1));
            ip = next_ip;
        }
    }
    const uint32_t enc = wave_readlane(wave_incl_sum_u32(enc_l), 63), ll0 = wave_readlane(wave_incl_sum_u32(ll0_l), 63);
    if (lane == 0) {
        strip[S_N * kCmpWaves + w] = nseq;
        strip[S_ENC * kCmpWaves + w] = enc;
        strip[S_LL0 * kCmpWaves + w] = ll0;
        strip[S_TAIL * kCmpWaves + w] = ce - anchor;
    }
}

// ------------------------------------------------------------------------------ phase 3, levels 10-12: optimal parse (one strip)
// What LZ4HC_compress_optimal does for the reference's levels 10-12 (lz4hc.c:1823-2130): instead of taking matches
// greedily with a two-position lookahead, choose the sequence boundaries that minimise the encoded size, given every
// position's longest match (any shorter length at the same offset may be used too).  price[c] = fewest bytes that encode
// the strip up to cell c; from cell p one can go to p + 1 by a literal (1 byte, 1 more when the literal run reaches a
// length-field boundary: 15, 270, ... lz4hc.c:1730-1740 LZ4HC_literalsPrice) or to p + ml by a match (token + offset +
// length bytes: 3, 4 from ml = 19 on; lz4hc.c:1743-1759 LZ4HC_sequencePrice).  A match longer than the window is taken at
// once, as the reference does with its sufficient length (lz4hc.c:1893-1901).
//
// Not the reference's loop: the prices of the next 64 cells live in ONE VGPR, cell p + 1 + l in lane l.  A step is a
// lane shift (DPP), one candidate per lane (lane l: the match of length l + 1 from p; lane 0: the literal) and a min -
// no memory access on the path from one position to the next.  The winning move of every cell (literal run length so
// far, or match length) goes to a 2-byte-per-position array (the search's second state array, free by now); a backward
// walk over it, staged through LDS in chunks, yields the sequences, which are recorded from the end of the strip's
// record area towards its start.
enum : uint32_t { kOptWin = 64, kOptLit = 0x8000u, kOptInf = 0xFFFFFF00u };

__device__ __forceinline__ void hc_parse_strip_opt(lz4amd_gsrc src, uint32_t n, const uint32_t* best_g, uint16_t* choice_g, MatchRec* recs, uint32_t rec_cap,
                                                   uint32_t* strip, uint32_t* first_rec, uint32_t* stage, uint32_t w, uint32_t cs, uint32_t ce, bool favor, uint64_t* prof = nullptr) {
    const uint32_t lane = lane_here();
    const uint64_t tf0 = prof ? clock_ticks() : 0;
    uint32_t nseq = 0, enc = 0, ll0 = 0, tail = ce - cs;
    if (n >= kMfLimit + 1 && cs <= n - kMfLimit) {
        const uint32_t last_q = n - kMfLimit;
        uint32_t mlimit = n - kLastLiterals; if (mlimit > ce) mlimit = ce;
        // ---- forward: prices
        uint32_t win = kOptInf;                                          // lane l: price << 8 | move of cell p + 1 + l (move: 0 literal, else match length)
        uint32_t Psh = 0, ll_cur = 0, ll_bump = 15;                      // cell p: its price << 8, the literal run that ends there; the run length at which the next length byte opens (15, 270, ...)
        // what a lane stands for: lane 0 the literal, lane l >= 3 the match of length l + 1 (lanes 1, 2: nothing)
        const uint32_t mlv = lane == 0 ? 0u : (lane < 3 ? 0xFFFFu : lane + 1);
        const uint32_t kv = lane == 0 ? (1u << 8) : (((lane + 1 >= 19 ? 4u : 3u) << 8) | (lane + 1));      // (price << 8 | move) a lane's step adds
        const uint32_t b0 = lane == 0 ? 256u : 0u;                       // ... plus this when the literal opens a new length byte
        uint32_t p = cs;
        while (p < ce) {
            // -- a group of 64 positions: their search results, cut to what may be used, one per lane; the moves of the group's cells
            const uint32_t gbase = p & ~63u, gend = gbase + 64 < ce ? gbase + 64 : ce;
            const uint32_t bv = best_g[gbase + lane];                   // (the array is padded)
            uint32_t Lv = bv & 0xFFu;
            {   const uint32_t pos = gbase + lane;
                if (favor && Lv > 18 && Lv <= 36) Lv = 18;               // lz4hc.c:1816-1818: no length byte for a few bytes more
                if (pos > last_q || Lv < kMinMatch || pos + kMinMatch > mlimit) Lv = 0;
                else if (Lv <= kOptWin && pos + Lv > mlimit) Lv = mlimit - pos; }
            uint32_t chv = 0;                                            // moves of cells [gbase, gbase + 64), lane = cell - gbase
            const uint32_t lo_cell = p + 1;                              // the first cell this pass notes
            bool jumped = false;
            while (p < gend) {
                const uint32_t Lc = wave_readlane(Lv, p & 63u);
                if (Lc > kOptWin) { jumped = true; break; }
                win = wave_next_u32(win);
                if (lane == 63) win = kOptInf;
                const uint32_t bump = ll_cur + 1 == ll_bump ? 1u : 0u;
                uint32_t cand = Psh + kv + (bump ? b0 : 0u);
                cand = mlv <= Lc ? cand : kOptInf;
                win = cand < win ? cand : win;
                const uint32_t head = wave_readlane(win, 0);
                const uint32_t move = head & 0xFFu;
                Psh = head & ~0xFFu;
                ll_cur = move ? 0u : ll_cur + 1;
                if (move) ll_bump = 15; else if (ll_cur == ll_bump) ll_bump += 255;
                p++;
                // the move of cell p (a literal cell notes its run length, at most 0x7FFF: longer runs chain); cell gbase + 64 is lane 0 of the next group
                const uint32_t val = move ? move : (kOptLit | (ll_cur < 0x7FFFu ? ll_cur : 0x7FFFu));
                if ((p & 63u) == 0) { if (lane == 0) choice_g[p] = (uint16_t)val; }
                else if (lane == (p & 63u)) chv = val;
            }
            {   // the group's cells noted in this pass: [lo_cell, p] without the cell that went to the next group
                const uint32_t cell = gbase + lane;
                if (cell >= lo_cell && cell <= p && cell < gbase + 64) choice_g[cell] = (uint16_t)chv;
            }
            if (jumped) {
                // -- a match longer than the window: taken at once (measured to its end if the search capped it)
                const uint32_t s1 = wave_readlane(bv, p & 63u);
                const uint32_t of = s1 >> 8;
                uint32_t ml = s1 & 0xFFu;
                if (ml >= kHcLenCap && p + ml < mlimit) {
                    for (;;) {
                        const uint32_t a = p + ml + 8 * lane;
                        uint32_t same = 0;
                        if (a < mlimit) { same = equal_bytes8(hc_ld8(src, n, a), hc_ld8(src, n, a - of)); if (same > mlimit - a) same = mlimit - a; }
                        const unsigned long long brk = __ballot(same < 8);
                        if (brk) { const uint32_t fl = (uint32_t)__ffsll((long long)brk) - 1; ml += 8 * fl + wave_readlane(same, fl); break; }
                        ml += 512;
                    }
                }
                if (p + ml > mlimit) ml = mlimit - p;
                if (ml > 0x7FFFu) ml = 0x7FFFu;                          // (the move is kept in 15 bits; the rest is found again from where this one ends)
                Psh += (3 + (ml >= 19 ? 1 + (ml - 19) / 255 : 0)) << 8;
                win = kOptInf; ll_cur = 0; ll_bump = 15;
                p += ml;
                if (lane == 0) choice_g[p] = (uint16_t)ml;
            }
        }
        if (prof && w == 0 && lane == 0) prof[3] += clock_ticks() - tf0;      // (developer profile: the forward pass of wave 0's strip)
        // (a literal run of more than 0x7FFF noted 0x7FFF in every cell behind that: the walk below takes it in steps)
        __threadfence_block();
        wave_lds_fence();
        // ---- backward: the sequences, from the strip's end (chunks of kHcChunk cells staged in LDS: moves and search results)
        uint16_t* cstage = (uint16_t*)stage;                             // moves of cells [sb, sb + kHcChunk]
        uint32_t* bstage = stage + kHcChunk;                             // search results of positions [sb, sb + kHcChunk)
        uint32_t sb = 0xFFFFFFFFu;
        auto stage_at = [&](uint32_t cell) {                             // make cell's chunk resident (cell > cs)
            const uint32_t want = cs + (((cell - 1 - cs) / kHcChunk) * kHcChunk);      // chunks hold cells (sb, sb + kHcChunk]
            if (want != sb) {
                sb = want;
                wave_lds_fence();
#pragma unroll
                for (uint32_t k = 0; k < kHcChunk / 64; k++) {
                    cstage[1 + lane + 64 * k] = choice_g[sb + 1 + lane + 64 * k];
                    bstage[lane + 64 * k] = best_g[sb + lane + 64 * k];
                }
                wave_lds_fence();
            }
        };
        uint32_t pos = ce;
        // trailing literals of the strip
        tail = 0;
        for (;;) {
            if (pos <= cs) break;
            stage_at(pos);
            const uint32_t c = cstage[pos - sb];
            if (!(c & kOptLit)) break;
            const uint32_t r = c & 0x7FFFu;
            tail += r; pos -= r;
        }
        // (the walk is one dependent chain of LDS reads: the move of the cell a match ends at is carried over from the step that
        //  found the cell, and a match's offset and the move of its first cell are read together)
        uint32_t ml = 0;
        if (pos > cs) { stage_at(pos); ml = cstage[pos - sb]; }         // a match ends at pos
        while (pos > cs) {
            const uint32_t start = pos - ml;
            uint32_t of, c1 = 0;
            if (start > sb) { of = bstage[start - sb] >> 8; c1 = cstage[start - sb]; }       // both in the staged chunk (the common case)
            else {
                of = best_g[start] >> 8;
                if (start > cs) { stage_at(start); c1 = cstage[start - sb]; }
            }
            // the literals before it: cell `start` says how many (0x7FFF: the run goes on below), then the cell a match ends at
            uint32_t ll = 0, q = start;
            while (q > cs && (c1 & kOptLit)) {
                const uint32_t r = c1 & 0x7FFFu;
                ll += r; q -= r;
                c1 = 0;
                if (q > cs) { stage_at(q); c1 = cstage[q - sb]; }
            }
            if (lane == 0) { MatchRec r; r.ll = ll; r.mo = of | ((ml - kMinMatch) << 16); recs[rec_cap - 1 - nseq] = r; }
            enc += enc_size(ll, ml - kMinMatch);
            ll0 = ll;
            nseq++;
            pos = q; ml = c1;                                            // (c1: the move of cell q, a match's end, or nothing at the origin)
        }
    }
    if (lane == 0) {
        strip[S_N * kCmpWaves + w] = nseq;
        strip[S_ENC * kCmpWaves + w] = enc;
        strip[S_LL0 * kCmpWaves + w] = ll0;
        strip[S_TAIL * kCmpWaves + w] = tail;
        first_rec[w] = rec_cap - nseq;                                   // where the strip's records start in its record area
    }
}

// ------------------------------------------------------------------------------ levels 1-2: two tables, one candidate each
// The reference's LZ4MID (lz4hc.c:472-773, k_clTable rows 0-2 lz4hc.c:93-95): no chains - one table keyed by a hash of 4 bytes and one
// keyed by a hash of 7 of the 8 bytes at a position (lz4hc.c:141-149: 2^14 entries each), each remembering the latest position, and a
// position's two candidates are what the tables held when it was reached ("long" candidate first, lz4hc.c:571-601; then the short one,
// 602-647).  Here, as at the other levels, linking and searching are separate passes over the block and EVERY position is linked and
// searched (the reference links a match's first and last positions only, lz4hc.c:688-711, and does not search inside a match): the
// link pass is hc_build_chain's with two head tables in the same 128 KB of LDS and leaves two distances per position; the search pass
// has no chain ring to hold, so the whole 64 KB window fits the LDS at once (hc_search_mid) - and there are no bands.  The lazy parse and the emit are phase 3 / 4 above (the reference's one-step look at ip + 1, lz4hc.c:618-634, is the lazy
// rule's first half).
enum : uint32_t { kMidHashLog = kHcHashLog - 1 };                        // lz4hc.c:141 LZ4MID_HASHLOG
__device__ __forceinline__ uint32_t mid_hash4(uint32_t v) { return (v * 2654435761u) >> (32 - kMidHashLog); }                          // lz4hc.c:144
__device__ __forceinline__ uint32_t mid_hash7(uint64_t v) { return (uint32_t)(((v << 8) * 58295818150454627ull) >> (64 - kMidHashLog)); }   // lz4hc.c:148 (the low 56 bits)

__device__ __forceinline__ void hc_build_links_mid(lz4amd_gsrc src, uint32_t n, uint16_t* d4_g, uint16_t* d8_g, char* smem) {
    const uint32_t tid = opaque_u32(threadIdx.x), lane = lane_here(), w = wave_id();
    uint32_t* misc = (uint32_t*)(smem + kHOffMisc);
    uint32_t* head4 = (uint32_t*)(smem + kHOffHead);
    uint32_t* head8 = head4 + (1u << kMidHashLog);
    uint32_t* wtab = (uint32_t*)(smem + kHOffWtab) + w * 256;
    for (uint32_t i = tid; i < (2u << kMidHashLog); i += kHcThreads) head4[i] = 0;
    for (uint32_t i = lane; i < 256; i += 64) wtab[i] = 0;
    if (tid == 0) misc[HM_TOKEN] = 0;
    __syncthreads();
    const uint32_t ngroups = (n + 63) / 64;
    const uint32_t nturns = (ngroups + kHcTurnGroups - 1) / kHcTurnGroups;
    for (uint32_t turn = w; turn < nturns; turn += kHcWaves) {
        uint32_t h4[kHcTurnGroups], h8[kHcTurnGroups], q4[kHcTurnGroups], q8[kHcTurnGroups];      // q: previous position + 1, 0 = none
        bool v4[kHcTurnGroups], v8[kHcTurnGroups], f4[kHcTurnGroups], f8[kHcTurnGroups], l4[kHcTurnGroups], l8[kHcTurnGroups];
#pragma unroll
        for (uint32_t j = 0; j < kHcTurnGroups; j++) {
            const uint32_t g0 = (turn * kHcTurnGroups + j) * 64, p = g0 + lane;
            v4[j] = p + 4 <= n; v8[j] = p + 8 <= n;                      // (lz4hc.c:548 ilimit: a position is keyed by 8 bytes while 8 are left)
            const uint64_t v = p < n ? hc_ld8(src, n, p) : 0;
            h4[j] = mid_hash4((uint32_t)v); h8[j] = mid_hash7(v);
            int pred;
            hc_group_links(h4[j], v4[j], wtab, lane, pred, l4[j]);
            f4[j] = pred < 0; q4[j] = pred >= 0 ? g0 + (uint32_t)pred + 1 : 0;
            hc_group_links(h8[j], v8[j], wtab, lane, pred, l8[j]);
            f8[j] = pred < 0; q8[j] = pred >= 0 ? g0 + (uint32_t)pred + 1 : 0;
        }
        while (lds_load_acquire(&misc[HM_TOKEN]) != turn) spin_pause();      // groups read and update the tables in position order
#pragma unroll
        for (uint32_t j = 0; j < kHcTurnGroups; j++) {
            const uint32_t p = (turn * kHcTurnGroups + j) * 64 + lane;
            if (v4[j] && f4[j]) q4[j] = head4[h4[j]];
            if (v8[j] && f8[j]) q8[j] = head8[h8[j]];
            wave_lds_order();
            if (v4[j] && l4[j]) head4[h4[j]] = p + 1;
            if (v8[j] && l8[j]) head8[h8[j]] = p + 1;
            wave_lds_order();
        }
        if (lane == 0) lds_store_relaxed(&misc[HM_TOKEN], turn + 1);
#pragma unroll
        for (uint32_t j = 0; j < kHcTurnGroups; j++) {
            const uint32_t p = (turn * kHcTurnGroups + j) * 64 + lane;
            uint32_t d4 = 0, d8 = 0;
            if (v4[j] && q4[j]) { const uint32_t d = p + 1 - q4[j]; if (d <= kMaxDistance) d4 = d; }
            if (v8[j] && q8[j]) { const uint32_t d = p + 1 - q8[j]; if (d <= kMaxDistance) d8 = d; }
            if (p < ngroups * 64) { d4_g[p] = (uint16_t)d4; d8_g[p] = (uint16_t)d8; }      // the arrays are padded to a multiple of 64 entries
        }
    }
    __syncthreads();
}

// The search: with no chain ring to hold, the LDS takes the whole LZ4 window - a 128 KB source ring, filled once per tile of 32 K
// positions: [t0 - 64 K, t0 + 32 K + kHcAhead) - and a position reads its own bytes and both candidates from it (reads of the
// candidates from L2 were tried first: one 16-byte read at an address of its own per lane ran at a fifth of this).  Four positions
// per lane and trip, so that their twelve reads are in flight together; only matches of sixteen bytes and more go round the
// measuring loop.
enum : uint32_t { kMidRing = 131072, kMidTile = 32768 };
static_assert(kHOffSrc + kMidRing + kHcPad <= kHcLdsBytes, "the two-table search's source ring must fit");
static_assert(kMidRing >= kMaxDistance + kMidTile + kHcAhead + 16, "the ring holds a tile, its window and the bytes a match is measured over");
__device__ __forceinline__ void mid_commit_src(uint8_t* ring, uint32_t P, const U32x4& v) {
    const uint32_t o = P & (kMidRing - 1);
    *(U32x4*)(ring + o) = v;
    if (o < kHcPad) *(U32x4*)(ring + kMidRing + o) = v;
}
// Matches of sixteen bytes and more.  The lanes of a wave hold consecutive positions, and a match at p over a distance d is a match
// one byte shorter at p + 1 over the same distance: so of a stretch of lanes whose first sixteen bytes all matched over one distance
// only the first (its `head`) is measured - by the whole wave, eight bytes a lane (30 lanes cover kHcLenCap), one step per head
// instead of a sixteen-byte loop that every lane of the wave sits through - and the others take the head's length less their
// distance to it.  p0: the position of lane 0; full: my first sixteen bytes equal those d bytes back.  Returns the length (not yet
// cut to the lane's own limit) for full lanes.
__device__ __forceinline__ uint32_t mid_extend(const uint8_t* ring, uint32_t p0, uint32_t d, bool full) {
    const uint32_t lane = lane_here();
    const uint32_t dkey = full ? d : 0u;                                 // (a candidate's distance is never 0)
    const uint32_t dprev = wave_prev_u32(dkey);                          // (every lane takes part in the shift)
    const bool head = full && dprev != dkey;
    const unsigned long long heads = __ballot(head);
    uint32_t lh = 16;
    for (unsigned long long rem = heads; rem; rem &= rem - 1) {
        const uint32_t f = (uint32_t)__ffsll((long long)rem) - 1;
        const uint32_t pf = p0 + f, df = wave_readlane(d, f);
        uint32_t same = 0;
        if (lane < 30) {                                                  // bytes [16 + 8 lane, 24 + 8 lane) of the two strings: up to 256, all inside the ring's kHcAhead
            const uint32_t o = (pf + 16 + 8 * lane) & (kMidRing - 1), q = (pf - df + 16 + 8 * lane) & (kMidRing - 1);
            const uint32_t* a = (const uint32_t*)(ring + (o & ~3u)); const uint32_t* b = (const uint32_t*)(ring + (q & ~3u));
            const uint32_t a0 = a[0], a1 = a[1], a2 = a[2], b0 = b[0], b1 = b[1], b2 = b[2];
            same = equal_bytes8_32(align_bytes(a1, a0, o & 3u), align_bytes(a2, a1, o & 3u), align_bytes(b1, b0, q & 3u), align_bytes(b2, b1, q & 3u));
        }
        const unsigned long long brk = __ballot(same < 8);               // (never empty: lanes 30.. report 0)
        const uint32_t fl = (uint32_t)__ffsll((long long)brk) - 1;
        const uint32_t len = 16 + 8 * fl + wave_readlane(same, fl);
        if (lane == f) lh = len;
    }
    // my head: the highest head lane at or below mine
    const unsigned long long upto = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1));
    const uint32_t hl = upto ? 63u - (uint32_t)__clzll((long long)upto) : lane;
    const uint32_t lhead = (uint32_t)__shfl((int)lh, (int)hl);
    const uint32_t gone = lane - hl;
    return lhead > gone + 16 ? lhead - gone : 16;                         // (a lane past its head's measured end still has its own sixteen)
}
__device__ __forceinline__ void hc_search_mid(lz4amd_gsrc src, uint32_t n, uint32_t first, const uint16_t* d4_g, const uint16_t* d8_g, uint32_t* st0_g, char* smem) {
    const uint32_t tid = opaque_u32(threadIdx.x);
    uint8_t* ring = (uint8_t*)(smem + kHOffSrc);
    const uint32_t n64 = (n + 63) & ~63u, last_q = n - kMfLimit;
    enum : uint32_t { K = 4 };
    uint32_t loaded = 0;                                                 // the ring holds [loaded - kMidRing, loaded)
    for (uint32_t t0 = 0; t0 < n64; t0 += kMidTile) {
        uint32_t H = t0 + kMidTile + kHcAhead; if (H > n) H = (n + 15) & ~15u;      // (kHcAhead is a multiple of 16; zero filled past n)
        __syncthreads();                                                 // nobody reads the bytes that are replaced any more
        for (uint32_t P = loaded + 16 * tid; P < H; P += 16 * kHcThreads) mid_commit_src(ring, P, load_src16(src, n, P));
        if (H > loaded) loaded = H;
        __syncthreads();
        uint32_t t1 = t0 + kMidTile; if (t1 > n64) t1 = n64;
        for (uint32_t base = t0; base < t1; base += K * kHcThreads) {
            uint32_t d8[K], d4[K];
            bool ok[K];
#pragma unroll
            for (uint32_t k = 0; k < K; k++) {
                const uint32_t p = base + k * kHcThreads + tid;
                ok[k] = p >= first && p <= last_q;
                d8[k] = ok[k] ? d8_g[p] : 0u; d4[k] = ok[k] ? d4_g[p] : 0u;
                if (d4[k] == d8[k]) d4[k] = 0;                           // one candidate
            }
            Q16 own[K], c8[K], c4[K];
#pragma unroll
            for (uint32_t k = 0; k < K; k++) {
                const uint32_t p = base + k * kHcThreads + tid;
                own[k] = lds_ld16(ring, p & (kMidRing - 1)); c8[k] = lds_ld16(ring, (p - d8[k]) & (kMidRing - 1)); c4[k] = lds_ld16(ring, (p - d4[k]) & (kMidRing - 1));
            }
#pragma unroll
            for (uint32_t k = 0; k < K; k++) {
                const uint32_t p = base + k * kHcThreads + tid;
                uint32_t lim = ok[k] ? n - kLastLiterals - p : 0u; if (lim > kHcLenCap) lim = kHcLenCap;
                uint32_t l8 = d8[k] ? equal_bytes16(own[k], c8[k]) : 0u, l4 = d4[k] ? equal_bytes16(own[k], c4[k]) : 0u;
                const bool full8 = l8 == 16, full4 = l4 == 16;
                if (__ballot(full8)) { const uint32_t l = mid_extend(ring, p - lane_here(), d8[k], full8); if (full8) l8 = l; }
                if (__ballot(full4)) { const uint32_t l = mid_extend(ring, p - lane_here(), d4[k], full4); if (full4) l4 = l; }
                if (l8 > lim) l8 = lim;
                if (l4 > lim) l4 = lim;
                uint32_t best = 0, off = 0;
                if (l8 >= kMinMatch) { best = l8; off = d8[k]; }
                if (l4 >= kMinMatch && l4 > best) { best = l4; off = d4[k]; }
                if (p < t1) st0_g[p] = best | (off << 8);
            }
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------ one block
__device__ __forceinline__ void hc_one_block(const HcBatch& P, uint32_t b, char* smem) {
    const uint32_t tid = opaque_u32(threadIdx.x), w = wave_id();
    uint32_t* misc = (uint32_t*)(smem + kHOffMisc);
    uint32_t* strip = (uint32_t*)(smem + kHOffStrip);
    const lz4amd_gsrc src0 = LZ4AMD_TO_GSRC(P.src[b]);
    const lz4amd_gdst dst = LZ4AMD_TO_GDST(P.dst[b]);
    const int32_t n_i = P.src_size[b];
    const int32_t cap_i = P.dst_cap[b];
    const lz4amd_gdst hints = P.hints ? LZ4AMD_TO_GDST(P.hints + (uint64_t)b * P.hint_stride) : (lz4amd_gdst)nullptr;      // optional entry-point table
    if (n_i < 0 || (uint32_t)n_i > kMaxInput || cap_i <= 0 || P.dst[b] == nullptr || (P.src[b] == nullptr && n_i != 0)) {
        if (tid == 0) { P.result[b] = 0; if (hints) *(uint32_t*)hints = 0; }                               // lz4hc.c:1403-1404
        return;
    }
    if (n_i == 0) { if (tid == 0) { dst[0] = 0; P.result[b] = 1; if (hints) *(uint32_t*)hints = 0; } return; }      // (no table for an empty block)
    // history (linked blocks / LZ4_compress_HC_continue in prefix mode, lz4hc.c:1666-1700): the `first` bytes right
    // before the block are linked into the chains like the block itself, but neither searched nor emitted.  From
    // here on positions count from the start of the history.  (A multiple of 64 keeps every per-position array aligned.)
    uint32_t first = P.prefix ? (uint32_t)P.prefix[b] : 0u;
    if (first > kMaxDistance + 1) first = kMaxDistance + 1;
    first &= ~63u;
    const lz4amd_gsrc src = src0 - first;
    const uint32_t n = (uint32_t)n_i + first, cap = (uint32_t)cap_i;
    uint8_t* scratch = P.scratch + (uint64_t)blockIdx.x * P.scratch_stride;
    uint16_t* chain_g = (uint16_t*)scratch;
    uint32_t* st0_g = (uint32_t*)(scratch + hc_chain_bytes(P.max_src));
    uint16_t* st1_g = (uint16_t*)(scratch + hc_chain_bytes(P.max_src) + hc_st0_bytes(P.max_src));
    MatchRec* recs_g = (MatchRec*)(scratch + hc_chain_bytes(P.max_src) + hc_st0_bytes(P.max_src) + hc_st1_bytes(P.max_src));
    HcEnt* list_g = (HcEnt*)(scratch + hc_chain_bytes(P.max_src) + hc_st0_bytes(P.max_src) + hc_st1_bytes(P.max_src) + hc_recs_bytes(P.max_src));
    uint32_t* count_g = (uint32_t*)(scratch + hc_chain_bytes(P.max_src) + hc_st0_bytes(P.max_src) + hc_st1_bytes(P.max_src) + hc_recs_bytes(P.max_src) + hc_list_bytes(P.max_src));
    uint64_t* prof = (LZ4AMD_HC_PROF && P.prof) ? P.prof + (uint64_t)blockIdx.x * 8 : nullptr;      // (a developer build's stamps: the product kernel carries none of that code)
    uint64_t tq = prof ? clock_ticks() : 0;

    uint32_t nstrips = 0, strip_len = 0;
    if (tid == 0) { misc[HM_OUT] = 0; misc[HM_CARRY] = 0; misc[HM_FAIL] = 0; }
    if ((uint32_t)n_i >= kMfLimit + 1) {
        // (P.level: the reference's compressionLevel in the low byte; LZ4AMD_HC_FAVOR_DEC_SPEED = 0x100 on top of it asks the optimal
        //  parse of levels 10-12 for the reference's decompression-speed preference, lz4hc.c:926-929, 1816-1818)
        const int level = P.level & 0xFF;
        const bool favor = (P.level & 0x100) != 0 && level >= 10;
        const bool mid = level >= 1 && level <= 2;                       // lz4hc.c:93-95 (a level below 1 is the default, 9: lz4hc.c:110-113)
        if (mid) hc_build_links_mid(src, n, chain_g, st1_g, smem); else hc_build_chain(src, n, chain_g, smem);
        if (prof && tid == 0) { const uint64_t t = clock_ticks(); prof[0] += t - tq; tq = t; }
        const uint32_t attempts = hc_attempts(level);

        if (!mid) for (uint32_t band = 0; band < kHcBands; band++) {
            // (the nearest band and the farther ones are two instances of the loop: what one of them never does is not in its code)
            if (band == 0) hc_search_band<true>(src, n, first, chain_g, st0_g, list_g, count_g, band, attempts, level >= 10 ? kHcSkipLenOpt : kHcSkipLenLazy, favor, smem, prof);
            else hc_search_band<false>(src, n, first, chain_g, st0_g, list_g, count_g, band, attempts, level >= 10 ? kHcSkipLenOpt : kHcSkipLenLazy, favor, smem, prof);
            if (prof && tid == 0) { const uint64_t t = clock_ticks(); prof[1 + (band ? 1 : 0)] += t - tq; tq = t; }
        } else {
            hc_search_mid(src, n, first, chain_g, st1_g, st0_g, smem);
            if (prof && tid == 0) { const uint64_t t = clock_ticks(); prof[1] += t - tq; tq = t; }
        }
        // -- parse: one wave per strip of the block proper
        const uint32_t own = n - first;
        nstrips = (own + kHcMinStrip - 1) / kHcMinStrip; if (nstrips > kHcWaves) nstrips = kHcWaves;
        strip_len = (((own + nstrips - 1) / nstrips) + 63) & ~63u;
        nstrips = (own + strip_len - 1) / strip_len;
        const uint32_t rec_cap = strip_len / 4 + 4;
        const bool optimal = level >= 10;                                // lz4hc.c:92-106: levels 10-12 are the optimal parser's
        if (w < nstrips) {
            const uint32_t cs = first + w * strip_len;
            uint32_t ce = cs + strip_len; if (ce > n) ce = n;
            if (optimal) hc_parse_strip_opt(src, n, st0_g, st1_g, recs_g + (uint64_t)w * rec_cap, rec_cap, strip, misc + HM_FIRST0, (uint32_t*)(smem + kHOffParse) + w * 2 * kHcChunk, w, cs, ce, favor, prof);
            else { hc_parse_strip(src, n, st0_g, recs_g + (uint64_t)w * rec_cap, strip, (uint32_t*)(smem + kHOffParse) + w * 2 * kHcChunk, w, cs, ce); if (lane_here() == 0) misc[HM_FIRST0 + w] = 0; }
        }
        __syncthreads();
        if (prof && tid == 0) { const uint64_t t = clock_ticks(); prof[4] += t - tq; tq = t; }
        // -- offsets (wave 0; limitedOutput: a block that does not fit fails as a whole, lz4hc.c:297-300)
        if (w == 0) {
            const StripTotals t = strip_offsets(strip, nstrips, 0, 0, 0, cap);
            // the strips' first sequence numbers, and the entry-point table's row distance: one row per 2^k sequences, about one per 512 bytes
            // of source (as lz4amd_k_compress chooses it tile by tile: settle_tile)
            const uint32_t l = lane_here();
            const uint32_t nk = l < nstrips ? strip[S_N * kCmpWaves + l] : 0u;
            const uint32_t n_incl = wave_incl_sum(nk);
            if (l < kCmpWaves) strip[6 * kCmpWaves + l] = n_incl - nk;
            if (l == 0) {
                misc[HM_OUT] = t.out; misc[HM_CARRY] = t.carry; misc[HM_FAIL] = t.fail; misc[HM_SEQS] = t.seqs; misc[HM_HOVER] = 0;
                const uint32_t want = (uint32_t)(((uint64_t)t.seqs << 9) / (own ? own : 1u));              // sequences per 512 bytes
                misc[HM_HK] = want >= LZ4AMD_HINT_EVERY_MAX ? LZ4AMD_HINT_EVERY_LOG2 : want >= 8 ? 3u : want >= 4 ? 2u : want >= 2 ? 1u : 0u;
            }
        }
        __syncthreads();
        // -- emit (the emit lanes also write the optional entry-point table's rows: lz4amd_params.h)
        if (w < nstrips && !misc[HM_FAIL] && strip[S_N * kCmpWaves + w]) {
            HintOut H; H.table = hints; H.cap_rows = LZ4AMD_HINT_CAP_ROWS(P.hint_stride); H.pre = first; H.over = &misc[HM_HOVER]; H.ord0 = 0; H.row0 = 0; H.k = misc[HM_HK];
            emit_strip(nullptr, recs_g + (uint64_t)w * rec_cap + misc[HM_FIRST0 + w], strip, w, src, dst, first + w * strip_len, 0xFFFFFFFFu, hints ? &H : nullptr, strip[6 * kCmpWaves + w]);
        }
        __syncthreads();
        if (prof && tid == 0) { const uint64_t t = clock_ticks(); prof[5] += t - tq; tq = t; }
    } else {
        __syncthreads();
        if (tid == 0) misc[HM_CARRY] = (uint32_t)n_i;
        __syncthreads();
    }
    // -- final literal run (lz4hc.c:1336-1357)
    const uint32_t out = misc[HM_OUT], run = misc[HM_CARRY];
    const uint64_t total = (uint64_t)out + 1 + lit_hdr_ext(run) + run;
    if (misc[HM_FAIL] || total > cap) { if (tid == 0) { P.result[b] = 0; if (hints) *(uint32_t*)hints = 0; } return; }
    if (hints && tid == 0) {
        // as lz4amd_k_compress ends its tables: the block's last sequence (its final literals) has a row of its own, the row behind it is the
        // block's end, row 0 carries the number of rows, the header makes the table valid
        const uint32_t cap_rows = LZ4AMD_HINT_CAP_ROWS(P.hint_stride);
        const uint32_t seqs = n > first && n - first >= kMfLimit + 1 ? misc[HM_SEQS] : 0u, k = seqs ? misc[HM_HK] : 0u;
        const uint32_t last_row = (seqs + (1u << k) - 1) >> k, nrows = last_row + 1;
        if (last_row && last_row < cap_rows) st_hint(hints, last_row, out, n - run - first, seqs);
        if (nrows <= cap_rows && !(seqs && misc[HM_HOVER]) && total < LZ4AMD_HINT_MAX_CSIZE) {
            hint_store_head(hints, (uint32_t)n_i, (uint32_t)total, seqs + 1, nrows);
        } else *(uint32_t*)hints = 0;
    }
    const uint32_t lit_dst = out + 1 + lit_hdr_ext(run);
    if (tid == 0) {
        lz4amd_gdst p = dst + out;
        if (run >= 15) { *p++ = 0xF0; put_len_ext(p, run - 15); }
        else *p++ = (uint8_t)(run << 4);
        P.result[b] = (int32_t)total;
    }
    const uint32_t sp = n - run;
    for (uint32_t i = tid; i < run; i += kHcThreads) dst[lit_dst + i] = src[sp + i];
}

// Workgroups pull blocks from a device-wide ticket counter.
__device__ __forceinline__ void hc_batch_body(const HcBatch& P) {
    LZ4AMD_DYN_LDS(smem);
    uint32_t* misc = (uint32_t*)(smem + kHOffMisc);
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) misc[HM_BLOCK] = take_ticket(P.ticket);
        __syncthreads();
        const uint32_t b = misc[HM_BLOCK];
        if (b >= P.n_blocks) break;
        hc_one_block(P, b, smem);
    }
}

} // namespace lz4amd
