// lz4_preparse_kernel.h -- stage A of the LZ4 block decoder (gfx950): the serial token chain of a block becomes a
// table of sequence records {output position, literal source, literal length, offset} in the workgroup's scratch.
// All 1024 threads of the workgroup take part; the stream is read from memory (L1 / L2), everything else is LDS.
//
// The stream is handled in SPANS of at most 1 MB (the span's token bitmap, one bit per stream byte, is 128 KB of LDS).
// A span is cut in up to 1024 SEGMENTS, one per thread:
//   P1  every thread walks the chain of its segment from the segment's first byte (the thread of the span's true
//       entry: from the entry), marking the token positions it visits in the bitmap.  A wrong start walks over
//       literal bytes misread as tokens (~6.5 B per step) and merges with the true chain after a few hundred bytes
//       (LZ4 chains self-synchronise).  One stream byte per trip, no divergence.
//   P2  every thread walks on from its exit (the "bridge") until it steps on a position a later thread marked.
//   P3  the true chain is stitched by induction: the entry's thread is true from the entry; where a true thread's
//       bridge merged into thread k's marks, thread k is true from there on.  The set of true threads is found by
//       pointer doubling over the merge links (10 rounds).  A bridge that never merged ends the span early at its
//       last (true) token - never a wrong answer, only a shorter span.
//   P4  marks before a thread's merge point and marks of skipped threads are dropped, the bridges of the true
//       threads are added: the bitmap now holds exactly the true tokens of the span.
//   P5  the token positions go to a list; 1024 tokens at a time, a thread decodes one sequence completely (both
//       length fields, offset), checks that it starts where its predecessor ended - the decoder, not the walk, is the
//       authority on the chain - and that it obeys the reference's input-side rules (read_variable_length
//       lz4.c:1979-2014); a block-wide scan places the sequences in the output; the output-side rules (lz4.c:2279,
//       2356, 2423) are applied and the records written, coalesced.
// Tokens with a length field longer than 64 bytes and the block's last sequence (lz4.c:2279, 2312-2318) go through a
// wave-cooperative SLOW PATH that takes one token at a time, whatever its size.
// A malformed block is rejected here, before a byte of output is written.
#pragma once
#include "lz4_common.h"
#include "../lz4amd_params.h"

namespace lz4amd { namespace pre {

struct alignas(16) SeqRec { uint32_t outpos, litpos, ll, off; };

enum : uint32_t {
    kThreads = 1024,
    kSpanMax = 1u << 20,                        // stream bytes per span
    kSegMin = 64,                              // a thread's segment is at least this long (short blocks use fewer threads; 256 -> 64: 64 KiB blocks decode 12 % faster)
    kBridgeTrips = 192,                         // lockstep trips of the bridge walk (96: more spans cut short, 4 MiB blocks 25 % slower; 384: no change)
    kBridgeFirst = 96,                          // ... of its first stage (32: the true chain is still walking more often than not; 64 / 128: as 96 within noise)
    kExtMax = 64,                               // longer length fields take the slow path
    kBias = 65536,                              // output positions are biased: [kBias - prefix, kBias) is the history before dst
    kNone = 0xFFFFFFFFu,
};

// LDS carve-up (bytes)
enum : uint32_t {
    kOffScan = 0,                               // u32[64] block scans
    kOffMisc = kOffScan + 64 * 4,               // u32[32]
    kOffOpos = kOffMisc + 32 * 4,               // u32[kThreads] where a thread's bridge ended
    kOffTin = kOffOpos + kThreads * 4,          // u32[kThreads] where the true chain enters a thread's marks
    kOffJump = kOffTin + kThreads * 4,          // u16[kThreads] merge links
    kOffKind = kOffJump + kThreads * 2,         // u8[kThreads] how a thread's bridge ended
    kOffMark = kOffKind + kThreads,             // u8[kThreads] on the true chain?
    kOffBitmap = (kOffMark + kThreads + 15) & ~15u,   // u32[kSpanMax / 32]
    kPreLdsBytes = kOffBitmap + kSpanMax / 8,
};
static_assert(kPreLdsBytes <= 152u * 1024u, "LDS budget");
enum : uint32_t { M_ERR = 1, M_MINREF = 2, M_TERM = 8, M_NX, M_OBASE, M_NREC, M_RC, M_HOVER, M_HROWS };     // M_HOVER / M_HROWS: entry-point table being made: a row did not fit / rows so far   // M_MINREF: lowest (biased) position any match reads

// scratch of one workgroup: the record table (every sequence but the last takes >= 3 stream bytes; +1 last, +1 sentinel)
// followed by the token list of one span
__host__ __device__ inline uint64_t table_bytes(uint32_t max_csize) { return ((uint64_t)max_csize / 3 + 4) * sizeof(SeqRec); }
__host__ __device__ inline uint64_t toks_bytes(uint32_t csize) {
    const uint32_t span = csize < kSpanMax ? csize : kSpanMax;
    return (((uint64_t)span / 3 + 8) * 4 + 15) & ~15ull;
}
// ... and behind the token list the REGION INDEX: for every 2 KB region of the block's output the record that holds the region's
// first byte (u32 per region): the copy stage of the decoder finds a region's records through it, straight from the table
#ifndef LZ4AMD_DEC_CHUNK
#define LZ4AMD_DEC_CHUNK 16
#endif
enum : uint32_t { kRegionShiftPre = LZ4AMD_DEC_CHUNK == 32 ? 11 : 10 };      // (lz4_decompress_kernel.h: a region is 64 chunks)
__host__ __device__ inline uint64_t max_regions(uint32_t max_csize, uint32_t max_out) {
    const uint64_t most = (uint64_t)max_csize * 255 + 64;                // the format's largest expansion
    const uint64_t out = max_out < most ? max_out : most;
    return (out >> kRegionShiftPre) + 4;
}
__host__ __device__ inline uint64_t scratch_bytes(uint32_t max_csize, uint32_t max_out) {
    return table_bytes(max_csize) + toks_bytes(max_csize) + max_regions(max_csize, max_out) * 4;
}

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
// 16 bytes of the compressed stream at position P (tail of the block zero padded; never reads past csize)
__device__ __forceinline__ U32x4 load_granule(lz4amd_gsrc src, uint32_t csize, uint32_t P) {
    if (P + 16 <= csize) return ld_global16(src + P);
    U32x4 v; v[0] = v[1] = v[2] = v[3] = 0;
#pragma nounroll
    for (uint32_t i = 0; i < 16 && P + i < csize; i++) chunk_set_byte(v, i, (uint32_t)src[P + i]);
    return v;
}

// A thread's view of the stream: 16 bytes in registers, reloaded when the position leaves them (one unaligned
// 16-byte load per ~2.5 walker steps instead of one byte load per step).  Measured and dropped: windows of 32 and of
// 64 bytes (a whole line asked for once) and refills in step across the wave - the walk's time did not move, the select
// trees of the wider windows made the bridging pass slower: a walker step is a chain of ~80 dependent instructions plus
// one load, and with four waves per SIMD that serial latency, not the memory system, sets the pace.
struct Win { U32x4 v; uint32_t base; };
__device__ __forceinline__ void win_init(Win& W) { W.v[0] = W.v[1] = W.v[2] = W.v[3] = 0; W.base = kNone - 64; }
__device__ __forceinline__ uint32_t win_byte(Win& W, lz4amd_gsrc g, uint32_t csize, uint32_t p) {      // p < csize
    if (p - W.base >= 16u) { W.base = p; W.v = load_granule(g, csize, p); }
    return chunk_byte(W.v, p - W.base);
}

// index in [from, 16) of the chunk's first byte that is not 255; 16 when there is none (from = 1 or 2).
// (the lowest set bit of ~chunk lies in the lowest byte that is not 255: no per-byte test needed)
__device__ __forceinline__ uint32_t first_not255(const U32x4& a, uint32_t from) {
    uint64_t lo = (uint64_t)(~a[0]) | ((uint64_t)(~a[1]) << 32), hi = (uint64_t)(~a[2]) | ((uint64_t)(~a[3]) << 32);
    lo &= ~0ull << (8 * from);
    if (lo) return ((uint32_t)__ffsll((long long)lo) - 1) >> 3;
    if (hi) return 8 + (((uint32_t)__ffsll((long long)hi) - 1) >> 3);
    return 16;
}

struct TokInfo { uint32_t ll, q, off, ml, nx, st; };      // st: 0 decoded, 1 slow path (last sequence, very long field), 2 malformed
// Decode the sequence whose token is at p (p < csize): the authority on the fields and on the reference's input-side rules.
__device__ __forceinline__ TokInfo tok_decode(lz4amd_gsrc g, uint32_t csize, uint32_t p) {
    TokInfo r; r.ll = 0; r.q = 0; r.off = 0; r.ml = 0; r.nx = 0; r.st = 1;
    const uint32_t b = g[p];
    uint32_t ll = b >> 4, q = p + 1;
    if (ll == 15) {
        uint32_t n = 0, x;
        do {
            if (n >= kExtMax) return r;
            if (q + 15 >= csize) { r.st = 2; return r; }              // lz4.c:1986-2006: a length byte is read only there
            x = g[q]; q++; n++; ll += x;
        } while (x == 255);
    }
    r.ll = ll; r.q = q;
    if (q > csize || csize - q < ll + 8) return r;                   // the block's last sequence (or a malformed one): slow path
    const uint32_t m = q + ll;
    uint32_t nx = m + 2, ml = b & 15;
    if (ml == 15) {
        uint32_t n = 0, x;
        do {
            if (n >= kExtMax) return r;
            x = g[nx]; nx++; n++; ml += x;
            if (nx + 4 > csize) { r.st = 2; return r; }
        } while (x == 255);
    }
    r.off = (uint32_t)g[m] | ((uint32_t)g[m + 1] << 8);
    r.ml = ml + kMinMatch; r.nx = nx; r.st = 0;
    return r;
}

// The walkers' view of the chain: one stream byte per trip, no branches.  A thread is at a token (mode 0), inside a
// literal-length field (mode 1) or inside a match-length field (mode 2).  It stops (dead, at the token `tok`) where
// tok_decode says "slow path" - and, off the true chain, wherever the bytes make no sense.
struct WalkState { uint32_t p, tok, acc, mode, cnt, mlf, dead; };       // (mlf, dead: 0 / 1 - loop-carried bools cost scalar mask merges every trip)
__device__ __forceinline__ void walk_init(WalkState& s, uint32_t p) { s.p = p; s.tok = p; s.acc = 0; s.mode = 0; s.cnt = 0; s.mlf = 0; s.dead = 0; }
// consume byte b = stream[s.p] (the caller checked s.p < csize)
__device__ __forceinline__ void walk_step(WalkState& s, uint32_t b, uint32_t csize) {
    const bool m0 = s.mode == 0, m1 = s.mode == 1, m2 = s.mode == 2;
    const bool is255 = b == 255;
    const uint32_t ll0 = b >> 4;
    const bool ext0 = ll0 == 15;
    const bool litdone = (m0 && !ext0) || (m1 && !is255);                  // the literal length is complete with this byte
    const uint32_t ll = m0 ? ll0 : s.acc + b;
    const uint32_t m = s.p + 1 + ll;                                        // first byte after the literals
    const bool mlf = m0 ? (b & 15) == 15 : s.mlf != 0;
    const bool cont = (m0 && ext0) || ((m1 || m2) && is255);               // the length field goes on
    const uint32_t cnt = m0 ? 0u : s.cnt + 1;
    const bool stop = (litdone && (m + 8 > csize || m < s.p)) || (cont && cnt >= kExtMax);
    s.tok = m0 ? s.p : s.tok;
    s.dead = stop ? 1u : 0u;
    s.acc = m0 ? 15u : s.acc + b;
    s.mlf = mlf ? 1u : 0u;
    s.cnt = litdone ? 0u : cnt;
    s.mode = litdone ? (mlf ? 2u : 0u) : (cont ? (m2 ? 2u : 1u) : 0u);
    s.p = litdone ? m + 2 : s.p + 1;
}

constexpr uint32_t kTokPerThread = 4;
// Where word w of the token bitmap lives in LDS.  Walker k marks and reads words around w = k * (segment / 32): with
// kilobyte segments that is a stride of 32 dwords - two banks for the whole wave, every LDS access of the walk serialised
// 32 ways (that, not memory, was what a walker step cost: ~2.3 K cycles per step of the workgroup).  Rotating every group
// of 32 words by its own number spreads neighbouring walkers over the banks.
__device__ __forceinline__ uint32_t bm_word(uint32_t w) { return (w & ~31u) | ((w + (w >> 5)) & 31u); }
enum : uint32_t { OUT_NONE = 0, OUT_MERGE = 1, OUT_EXIT = 2, OUT_STOP = 3, OUT_IDLE = 4 };

// SLOW PATH (wave 0, every lane the same values; length fields are scanned 64 bytes at a time): the sequence whose
// token is at p, whatever its size, with the reference's rules; its record goes to rectab[nrec].
// Returns 0: go on at nx, 1: that was the block's last sequence, 2: malformed (position in errpos).
// The region index: record rec covers output [o, oe); if the first byte of a 1 KB region lies in there, the FIRST such
// region notes rec + 1 (0 = nothing noted: the index is zeroed before, and a sequence that covers several region
// starts leaves holes behind the first, which the reader fills with a running maximum - entries only ever grow)
__device__ __forceinline__ void region_note(uint32_t* ridx, uint32_t o, uint32_t oe, uint32_t rec, bool on) {
    const uint32_t g = (o + (1u << kRegionShiftPre) - 1) >> kRegionShiftPre;
    if (on && (g << kRegionShiftPre) < oe) ridx[g - (kBias >> kRegionShiftPre)] = rec + 1;
}

__device__ __forceinline__ int slow_token(lz4amd_gsrc g, uint32_t csize, uint32_t capB, uint32_t low, uint32_t p,
                                          uint32_t& obase, uint32_t& nrec, SeqRec* rectab, uint32_t* ridx, uint32_t& nx_out, uint32_t& errpos, uint32_t& minref) {
    const uint32_t lane = lane_here();
    errpos = p < csize ? p : (csize ? csize - 1 : 0);
    if (p >= csize) return 2;
    const uint32_t t = g[p];
    uint32_t ll = t >> 4, q = p + 1;
    if (ll == 15) {
        for (;;) {
            const uint32_t pos = q + lane;
            const bool inb = pos + 15 < csize;
            const uint32_t x = inb ? (uint32_t)g[pos] : 0u;
            const unsigned long long stopm = __ballot(!inb || x != 255);
            if (!stopm) { ll += 255 * 64; q += 64; if (ll > csize) return 2; continue; }
            const uint32_t kk = (uint32_t)__ffsll((long long)stopm) - 1;
            if (!wave_readlane(inb ? 1u : 0u, kk)) return 2;
            ll += 255 * kk + wave_readlane(x, kk); q += kk + 1;
            break;
        }
        if (ll > csize) return 2;
    }
    const uint32_t rem = csize - q, room = capB - obase;
    const bool last = rem < ll + 8 || room < ll + kMfLimit;                 // lz4.c:2279
    if (last) {
        if (rem != ll || room < ll) return 2;                               // lz4.c:2312-2318
        if (lane == 0) { SeqRec r; r.outpos = obase; r.litpos = q; r.ll = ll; r.off = 0; rectab[nrec] = r; }
        region_note(ridx, obase, obase + ll, nrec, lane == 0);
        nrec++; obase += ll;
        return 1;
    }
    const uint32_t m = q + ll;                                              // m + 8 <= csize
    const uint32_t off = (uint32_t)g[m] | ((uint32_t)g[m + 1] << 8);
    uint32_t ml = t & 15, nx = m + 2;
    if (ml == 15) {
        for (;;) {
            const uint32_t pos = nx + lane;
            const uint32_t x = pos < csize ? (uint32_t)g[pos] : 0u;
            const bool badafter = pos + 5 > csize;                          // after a length byte at least 4 more bytes must follow
            const unsigned long long stopm = __ballot(x != 255 || badafter);
            if (!stopm) { ml += 255 * 64; nx += 64; if (ml > 0x7FFFFFF0u) return 2; continue; }
            const uint32_t kk = (uint32_t)__ffsll((long long)stopm) - 1;
            ml += 255 * kk + wave_readlane(x, kk); nx += kk + 1;
            if (wave_readlane(badafter ? 1u : 0u, kk) || ml > 0x7FFFFFF0u) return 2;
            break;
        }
    }
    ml += kMinMatch;
    const uint32_t ms = obase + ll;
    if (off == 0 || off > ms - low) return 2;                               // lz4.c:2356
    if (ms - off < minref) minref = ms - off;
    if (capB - ms < ml + kLastLiterals) return 2;                           // lz4.c:2423
    if (lane == 0) { SeqRec r; r.outpos = obase; r.litpos = q; r.ll = ll; r.off = off; rectab[nrec] = r; }
    region_note(ridx, obase, ms + ml, nrec, lane == 0);
    nrec++; obase = ms + ml;
    nx_out = nx;
    return 0;
}

// ------------------------------------------------------------------------------ stage A
// Returns false (uniformly) when the block is malformed (position in misc[M_ERR]); nseq_out / total_out otherwise.
// hint_out (optional): the block's entry-point table is written on the way (include/lz4amd.h; layout csrc/lz4amd_params.h) - a row
// per 2^k tokens, k <= 3 set per trip of P5 so that rows are ~512 bytes of output apart, one row per token of the slow path - so
// that the next decode of the same block can skip this stage; hint_cap_rows: rows the table has room for.
__device__ __forceinline__ bool preparse_block(lz4amd_gsrc src, uint32_t csize, uint32_t cap, uint32_t prefix,
                                               SeqRec* rectab, char* smem, uint64_t table_size,
                                               uint32_t& nseq_out, uint32_t& total_out, uint64_t* prof = nullptr, uint32_t** ridx_out = nullptr,
                                               lz4amd_gdst hint_out = nullptr, uint32_t hint_cap_rows = 0) {
    uint64_t pt[6] = {0, 0, 0, 0, 0, 0}, pq = prof ? clock_ticks() : 0;       // developer profile: cycles in P1, P2, P3, P4, list, P5
#define LZ4AMD_PSTAMP(i) do { if (prof) { const uint64_t t_ = clock_ticks(); pt[i] += t_ - pq; pq = t_; } } while (0)
    const uint32_t tid = threadIdx.x;
    uint32_t* scan = (uint32_t*)(smem + kOffScan);
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    uint32_t* oposv = (uint32_t*)(smem + kOffOpos);
    uint32_t* tinv = (uint32_t*)(smem + kOffTin);
    uint16_t* jump = (uint16_t*)(smem + kOffJump);
    uint8_t* kindv = (uint8_t*)(smem + kOffKind);
    uint8_t* mark = (uint8_t*)(smem + kOffMark);
    uint32_t* bm = (uint32_t*)(smem + kOffBitmap);
    uint32_t* toks = (uint32_t*)((char*)rectab + table_size);
    uint32_t* ridx = (uint32_t*)((char*)toks + toks_bytes(csize));       // region (from the block's first) -> record
    if (ridx_out) *ridx_out = ridx;
    const uint32_t capB = cap + kBias, low = kBias - prefix;
    if (tid == 0) { misc[M_ERR] = kNone; misc[M_MINREF] = kNone; misc[M_HOVER] = 0; }
    uint32_t hrows = 0;                               // uniform: rows of the entry-point table written so far
    for (uint32_t i = tid, n = (uint32_t)max_regions(csize, cap); i < n; i += kThreads) ridx[i] = 0;    // (ordered before P5's notes by the barriers between)

    uint32_t e = 0, obase = kBias, nrec = 0;          // uniform: next true token, output position, records written
    for (;;) {
        // ---- the span and its segments
        const uint32_t sp0 = e & ~31u;
        const uint32_t span = csize - sp0 < kSpanMax ? csize - sp0 : kSpanMax, sp1 = sp0 + span;
        uint32_t S = ((span + kThreads - 1) / kThreads + 63) & ~63u; if (S < kSegMin) S = kSegMin;
        const uint32_t nl = (span + S - 1) / S, wps = S / 32;           // threads in use, bitmap words per segment
        const uint32_t je = (e - sp0) / S;
        for (uint32_t w = tid; w < (((span + 31) / 32 + 31) & ~31u); w += kThreads) bm[w] = 0;      // whole 32-word groups (bm_word)
        __syncthreads();
        // ---- P1: walk my segment, marking the tokens
        const uint32_t seg_lo = sp0 + tid * S, seg_hi = seg_lo + S;
        const bool inuse = tid < nl && tid >= je;
        WalkState s; walk_init(s, tid == je ? e : seg_lo);
        Win W; win_init(W);
        // (straight-line with selects: the nested-if form of this loop compiled to twice the instructions, most of them
        //  scalar mask bookkeeping, and a walker step is latency bound by its own instruction chain)
        for (;;) {
            const bool at_tok = s.mode == 0;
            const bool act = inuse && !s.dead && !(at_tok && s.p >= seg_hi);
            if (!__any(act)) break;
            const bool off = s.p >= csize;                                   // ran off the block
            const bool go = act && !off;
            if (go && at_tok) atomicOr(&bm[bm_word((s.p - sp0) >> 5)], 1u << ((s.p - sp0) & 31));
            if (go && s.p - W.base >= 16u) { W.base = s.p; W.v = load_granule(src, csize, s.p); }
            WalkState t = s;
            walk_step(t, chunk_byte(W.v, (s.p - W.base) & 15u), csize);
            s.tok = go ? t.tok : (act && at_tok ? s.p : s.tok);
            s.dead = go ? t.dead : (s.dead | (act ? 1u : 0u));
            s.acc = go ? t.acc : s.acc; s.mlf = go ? t.mlf : s.mlf; s.cnt = go ? t.cnt : s.cnt;
            s.mode = go ? t.mode : s.mode; s.p = go ? t.p : s.p;
        }
        __syncthreads();
        LZ4AMD_PSTAMP(0);
        // ---- P2: walk on until a marked position (a token of a later thread's walk)
        const uint32_t x = s.p;
        uint32_t okind = !inuse ? OUT_IDLE : (s.dead ? OUT_STOP : OUT_NONE), opos = s.dead ? s.tok : s.p, nb = 0;
        // (in stages: all that matters is that the threads ON THE TRUE CHAIN have merged, and they do so within a few tokens; the
        //  slowest of a thousand walkers - one that started inside literals and decodes noise - nearly always needs all the trips
        //  there are, and every wave would wait for it.  So: a short walk, P3; only if the true chain ends in a thread that is
        //  still walking, the walk goes on.)
        uint32_t trip = 0, stage_limit = kBridgeFirst;
        bool active; uint32_t term, tend, myT; bool stop;
        for (;;) {
        for (;; trip++) {
            const bool run = okind == OUT_NONE;
            if (!__any(run)) break;
            if (trip >= stage_limit) break;
            const bool at_tok = s.mode == 0;
            const bool c_exit = at_tok && s.p >= sp1;
            uint32_t bit = 0;
            if (run && at_tok && !c_exit) bit = (bm[bm_word((s.p - sp0) >> 5)] >> ((s.p - sp0) & 31)) & 1u;
            const bool c_merge = bit != 0;
            const bool c_trips = trip >= kBridgeTrips;                       // (a token either way: the next span starts there)
            const bool c_end = s.p >= csize;
            const bool go = run && !c_exit && !c_merge && !c_trips && !c_end;
            if (go && s.p - W.base >= 16u) { W.base = s.p; W.v = load_granule(src, csize, s.p); }
            WalkState t = s;
            walk_step(t, chunk_byte(W.v, (s.p - W.base) & 15u), csize);
            const uint32_t here = at_tok ? s.p : s.tok;
            const uint32_t kind_stop = c_exit ? OUT_EXIT : c_merge ? OUT_MERGE : c_trips ? OUT_EXIT : OUT_STOP;
            const uint32_t pos_stop = (c_exit || c_merge) ? s.p : here;
            okind = !run ? okind : (go ? (t.dead ? OUT_STOP : OUT_NONE) : kind_stop);
            opos = !run ? opos : (go ? (t.dead ? t.tok : opos) : pos_stop);
            nb += go && at_tok ? 1u : 0u;
            s.tok = go ? t.tok : s.tok; s.dead = go ? t.dead : s.dead; s.acc = go ? t.acc : s.acc; s.mlf = go ? t.mlf : s.mlf;
            s.cnt = go ? t.cnt : s.cnt; s.mode = go ? t.mode : s.mode; s.p = go ? t.p : s.p;
        }
        // ---- P3: which threads are on the true chain?  (merge links, pointer doubling)
        __syncthreads();
        LZ4AMD_PSTAMP(1);
        oposv[tid] = opos; kindv[tid] = (uint8_t)okind;
        jump[tid] = (uint16_t)(okind == OUT_MERGE ? (opos - sp0) / S : tid);
        mark[tid] = tid == je ? 1 : 0;
        tinv[tid] = tid == je ? e : kNone;
        __syncthreads();
        for (uint32_t r = 0; r < 10; r++) {
            const uint32_t jt = jump[tid];
            const bool mine = mark[tid] != 0;
            __syncthreads();
            if (mine) mark[jt] = 1;
            const uint32_t j2 = jump[jt];
            __syncthreads();
            jump[tid] = (uint16_t)j2;
            __syncthreads();
        }
        active = mark[tid] != 0;
        if (active && okind == OUT_MERGE) tinv[(opos - sp0) / S] = opos;
        if (active && okind != OUT_MERGE) misc[M_TERM] = tid;
        __syncthreads();
        term = misc[M_TERM];
        tend = oposv[term];
        stop = kindv[term] == OUT_STOP;
        myT = tinv[tid];
        LZ4AMD_PSTAMP(2);
        if (kindv[term] != OUT_NONE || stage_limit > kBridgeTrips) break;       // (the second stage runs to the end: c_trips)
        stage_limit = kBridgeTrips + 1;
        __syncthreads();
        }
        // ---- P4: the bitmap of the true tokens in [e, tend)
        for (uint32_t w = 0; w < wps; w++) {
            const uint32_t base = seg_lo + 32 * w;
            if (tid < nl && base < sp1) {
                uint32_t v = bm[bm_word(tid * wps + w)];
                if (!active) v = 0;
                else if (myT > base) v = (myT - base >= 32) ? 0u : (v & (0xFFFFFFFFu << (myT - base)));
                if (tend <= base) v = 0; else if (tend - base < 32) v &= (1u << (tend - base)) - 1u;
                bm[bm_word(tid * wps + w)] = v;
            }
        }
        __syncthreads();
        {   // the bridges of the true threads (walked again: they were not kept)
            WalkState b; walk_init(b, x);
            win_init(W);
            uint32_t left = active ? nb : 0;
            while (__any(left != 0)) {
                const bool at_tok = b.mode == 0;
                if (left && at_tok && b.p < tend && b.p < sp1) atomicOr(&bm[bm_word((b.p - sp0) >> 5)], 1u << ((b.p - sp0) & 31));
                left -= left && at_tok ? 1u : 0u;
                const bool go = left != 0;
                if (go && b.p - W.base >= 16u) { W.base = b.p; W.v = load_granule(src, csize, b.p); }
                WalkState t = b;
                walk_step(t, chunk_byte(W.v, (b.p - W.base) & 15u), csize);
                b.tok = go ? t.tok : b.tok; b.dead = go ? t.dead : b.dead; b.acc = go ? t.acc : b.acc; b.mlf = go ? t.mlf : b.mlf;
                b.cnt = go ? t.cnt : b.cnt; b.mode = go ? t.mode : b.mode; b.p = go ? t.p : b.p;
            }
        }
        __syncthreads();
        LZ4AMD_PSTAMP(3);
        // ---- the token list
        uint32_t cnt = 0;
        for (uint32_t w = 0; w < wps; w++) if (tid < nl && seg_lo + 32 * w < sp1) cnt += (uint32_t)__popc(bm[bm_word(tid * wps + w)]);
        uint32_t ea, ta; uint64_t eb, tb;
        block_excl_sum2(cnt, 0ull, scan, ea, eb, ta, tb);
        {
            uint32_t k = ea;
            for (uint32_t w = 0; w < wps; w++) {
                if (tid < nl && seg_lo + 32 * w < sp1) {
                    uint32_t v = bm[bm_word(tid * wps + w)];
                    while (v) { const uint32_t b = (uint32_t)__ffs((int)v) - 1; v &= v - 1; toks[k++] = seg_lo + 32 * w + b; }
                }
            }
        }
        const uint32_t N = ta;
        __syncthreads();
        LZ4AMD_PSTAMP(4);
        // ---- P5: decode, check, place, write the records - 4 consecutive tokens per thread, 4096 per trip.  A token costs
        // two dependent trips to memory (its own bytes, then the offset / match length behind its literals); a thread
        // keeps the four of them in flight together (one token per thread and trip was bound by exactly that latency:
        // 0.7 M cycles per 4 MiB block).  The fast form below covers length fields that end inside the 16 bytes loaded (lengths < 3585), away
        // from the block's end; anything else goes through tok_decode, the authority.
        int bad = 0;
        for (uint32_t base = 0; base < N; base += kThreads * kTokPerThread) {
            // wave w takes the tokens base + 256 w .. + 255; lane l the tokens l, l + 64, l + 128, l + 192 of them, so that
            // every load and store of the wave walks over neighbouring tokens (a lane with four tokens in a row touched
            // 64 different cache lines per instruction, and the pass was bound by that, not by latency)
            const uint32_t i0 = base + (tid >> 6) * (64 * kTokPerThread) + (tid & 63);
            uint32_t tp[kTokPerThread], tnext[kTokPerThread];
#pragma unroll
            for (uint32_t j = 0; j < kTokPerThread; j++) {
                const uint32_t i = i0 + 64 * j;
                tp[j] = i < N ? toks[i] : tend;
                tnext[j] = i + 1 < N ? toks[i + 1] : tend;
            }
            TokInfo ti[kTokPerThread];
            U32x4 g0[kTokPerThread], g1[kTokPerThread];
            bool fast[kTokPerThread];
            uint32_t mpos[kTokPerThread];
#pragma unroll
            for (uint32_t j = 0; j < kTokPerThread; j++) {
                fast[j] = i0 + 64 * j < N && tp[j] + 16 <= csize;
                g0[j][0] = g0[j][1] = g0[j][2] = g0[j][3] = 0;
                if (fast[j]) g0[j] = ld_global16(src + tp[j]);
            }
#pragma unroll
            for (uint32_t j = 0; j < kTokPerThread; j++) {
                const uint32_t b = g0[j][0] & 0xFFu;
                const bool ext = (b >> 4) == 15;
                const uint32_t k = first_not255(g0[j], 1);                     // 15 + 255 per full byte + the byte that ends the field
                ti[j].ll = ext ? 15 + 255 * (k - 1) + chunk_byte(g0[j], k & 15) : b >> 4;
                ti[j].q = tp[j] + (ext ? k + 1 : 1u);
                mpos[j] = ti[j].q + ti[j].ll;
                fast[j] = fast[j] && !(ext && k == 16) && mpos[j] + 16 <= csize;
                g1[j][0] = g1[j][1] = g1[j][2] = g1[j][3] = 0;
                if (fast[j]) g1[j] = ld_global16(src + mpos[j]);
            }
#pragma unroll
            for (uint32_t j = 0; j < kTokPerThread; j++) {
                const uint32_t b = g0[j][0] & 0xFFu;
                const bool ext = (b & 15) == 15;
                const uint32_t k = first_not255(g1[j], 2);
                const uint32_t nx = mpos[j] + (ext ? k + 1 : 2u);
                if (fast[j] && !(ext && k == 16) && nx + 4 <= csize) {
                    ti[j].off = g1[j][0] & 0xFFFFu;
                    ti[j].ml = (ext ? 15 + 255 * (k - 2) + chunk_byte(g1[j], k & 15) : b & 15) + kMinMatch;
                    ti[j].nx = nx;
                    ti[j].st = 0;
                } else if (i0 + 64 * j < N) ti[j] = tok_decode(src, csize, tp[j]);
                else { ti[j].ll = ti[j].q = ti[j].off = ti[j].ml = ti[j].nx = 0; ti[j].st = 0; }
            }
            // output positions: a scan over the wave's 256 tokens in token order (lane fastest), then over the waves
            uint32_t excl[kTokPerThread], wsum = 0;             // 256 sequences of < 2^16 bytes each (fields of <= kExtMax bytes): 32 bits are plenty inside the wave
#pragma unroll
            for (uint32_t j = 0; j < kTokPerThread; j++) {
                const uint32_t len = i0 + 64 * j < N && ti[j].st == 0 ? ti[j].ll + ti[j].ml : 0u;
                const uint32_t incl = wave_incl_sum(len);
                excl[j] = wsum + incl - len;
                wsum += wave_readlane(incl, 63);
            }
            uint32_t e2, t2; uint64_t eo, to;
            block_excl_sum2(0u, (tid & 63) == 0 ? (uint64_t)wsum : 0ull, scan, e2, eo, t2, to);
            const uint64_t wbase = (uint64_t)obase + wave_readlane64(eo, 0);
            // the trip's rows of the table: ~512 bytes of output apart, at most 8 tokens
            const uint32_t ntrip = N - base < kThreads * kTokPerThread ? N - base : kThreads * kTokPerThread;
            uint32_t hk = 3;
            if (hint_out) { const uint32_t tpr = to ? (uint32_t)(((uint64_t)ntrip << 9) / to) : 8u; hk = tpr >= 8 ? 3u : tpr >= 4 ? 2u : tpr >= 2 ? 1u : 0u; }
#pragma unroll
            for (uint32_t j = 0; j < kTokPerThread; j++) {
                if (i0 + 64 * j < N) {
                    const uint64_t o64 = wbase + excl[j];
                    const uint32_t o = (uint32_t)o64;
                    bool b = ti[j].st != 0 || ti[j].nx != tnext[j] || (i0 + 64 * j == 0 && tp[j] != e);   // the chain, as the decoder sees it
                    b = b || o64 > capB || capB - o < ti[j].ll + kMfLimit || ti[j].off == 0 || ti[j].off > o + ti[j].ll - low
                          || capB - (o + ti[j].ll) < ti[j].ml + kLastLiterals;          // lz4.c:2279 (a sequence here is never the last), 2356, 2423
                    if (b) { atomicMin(&misc[M_ERR], tp[j]); bad = 1; }
                    else if (o + ti[j].ll - ti[j].off < kBias) atomicMin(&misc[M_MINREF], o + ti[j].ll - ti[j].off);    // reaches into the history
                    if (!b) {
                        SeqRec r; r.outpos = o; r.litpos = ti[j].q; r.ll = ti[j].ll; r.off = ti[j].off; rectab[nrec + i0 + 64 * j] = r;
                        region_note(ridx, o, o + ti[j].ll + ti[j].ml, nrec + i0 + 64 * j, true);
                        const uint32_t x = i0 + 64 * j - base;
                        if (hint_out && (x & ((1u << hk) - 1u)) == 0) {
                            const uint32_t row = hrows + (x >> hk);
                            if (row >= hint_cap_rows) misc[M_HOVER] = 1u;
                            else if (row) hint_store_row(hint_out, row, tp[j], o - kBias, nrec + i0 + 64 * j);      // (row 0 is written at the end: it carries the number of rows)
                        }
                    }
                }
            }
            if (__syncthreads_or(bad)) return false;
            obase += (uint32_t)to;
            hrows += (ntrip + (1u << hk) - 1u) >> hk;
        }
        nrec += N;
        LZ4AMD_PSTAMP(5);
        // ---- a token the walk could not pass: the slow path takes it (wave 0), then the next span starts behind it
        if (stop) {
            if (hint_out && tid == 0) {             // (the slow path's token gets a row of its own)
                if (hrows >= hint_cap_rows) misc[M_HOVER] = 1u;
                else if (hrows) hint_store_row(hint_out, hrows, tend, obase - kBias, nrec);
            }
            hrows++;
            if (tid < 64) {
                uint32_t nx = 0, ep = 0, mr = kNone;
                const int rc = slow_token(src, csize, capB, low, tend, obase, nrec, rectab, ridx, nx, ep, mr);
                if (tid == 0 && mr < misc[M_MINREF]) misc[M_MINREF] = mr;
                if (tid == 0) { misc[M_RC] = (uint32_t)rc; misc[M_NX] = nx; misc[M_OBASE] = obase; misc[M_NREC] = nrec; if (rc == 2) misc[M_ERR] = ep; }
            }
            __syncthreads();
            const uint32_t rc = misc[M_RC];
            if (rc == 2) return false;
            obase = misc[M_OBASE]; nrec = misc[M_NREC];
            if (rc == 1) break;
            e = misc[M_NX];
        } else e = tend;
        __syncthreads();
    }
    if (tid == 0) { SeqRec r; r.outpos = obase; r.litpos = csize; r.ll = 0; r.off = 0; rectab[nrec] = r; }      // sentinel row
    nseq_out = nrec; total_out = obase - kBias;
    if (hint_out) {
        __syncthreads();                                      // (M_HOVER)
        if (tid == 0) {
            if (!misc[M_HOVER] && hrows && hrows <= hint_cap_rows && csize < LZ4AMD_HINT_MAX_CSIZE)
                hint_store_head(hint_out, obase - kBias, csize, nrec, hrows);      // the block's end, the first sequence, the number of rows: the table is valid
        }
    }
    if (prof && tid == 0) { prof[2] = pt[0] | (pt[1] << 32); prof[3] = pt[2] | (pt[3] << 32); prof[4] = pt[4] | (pt[5] << 32); }
#undef LZ4AMD_PSTAMP
    return true;
}

} } // namespace lz4amd::pre
