// lz4_decompress_kernel.h -- batched LZ4 block decompression for gfx950 (MI355X).
//
// Replaces, for a whole batch of independent blocks resident in HBM, what the reference does per
// block in LZ4_decompress_safe (lib/lz4.c:2451 -> LZ4_decompress_generic lz4.c:2023-2445; length
// fields: read_variable_length lz4.c:1979-2014; end-of-block rules lz4.c:2276-2330, 2421-2429).
// Accepts ANY legal LZ4 block, rejects what the reference's safe loop rejects, never writes outside
// dst[0,cap) and never uses a byte outside src[0,csize) (the stream stage fetches the block in the
// aligned 16-byte granules of its memory: the first and the last granule may reach up to 15 bytes
// beyond the block, inside the granule, hence the page, that holds its first / last byte).  Not a port: the reference decoder is one serial
// token chain per block.  Here ONE 1024-thread workgroup (16 waves, one CU, ~150 KB of its LDS)
// decodes a block in two stages:
//
//   A PRE-PARSE  (all 16 waves, lz4_preparse_kernel.h) turns the serial token chain into a table of
//       sequence records {output position, literal source, literal length, offset} in the
//       workgroup's scratch, applying every format rule on the way: a malformed block is rejected
//       before a byte of output is written.  A block that comes with an ENTRY-POINT TABLE
//       (include/lz4amd.h: rows that name every few sequences of the chain, written by
//       lz4amd_k_compress or anybody else) skips this stage: see PARSER below.  On request (hint_make) the
//       stage writes that table for a block that came without a usable one, for the block's next decode.
//
//   B STREAM     a dataflow pipeline through LDS rings and counters only - no workgroup barrier
//       between the first and the last byte of the block:
//       MOVER (wave 15)  moves everything the others read into LDS by LDS-DMA (global_load_lds_dwordx4:
//           1 KB per instruction, no registers in between, up to 16 KB in flight) and does nothing
//           else: the compressed block -> a 32 KB ring, in the aligned 16-byte granules of its memory;
//           without a table the record table -> a 2048-row ring and the pre-parse's region index (for
//           every 1 KB REGION of output the record that holds its first byte) -> a 512-entry ring; with
//           a table its rows -> a 256-row ring.  It also moves the first-open-region word over the
//           complete marks of the regions in flight.
//       PARSER (waves 13-14, blocks with a table)  claim batches of rows under a lock, walk them - lane =
//           row, every lane the sequences up to the next row, out of the compressed ring, with the
//           pre-parse's rules - write records and region index straight into the rings and publish in
//           claim order: a row is trusted only because the lane before it arrived exactly there.
//       COPY (the other waves)  output-stationary: regions are handed out in order to whichever wave is
//           free.  A region is composed in its slot of a 80 KB LDS ring that always holds the 64 KB LZ4
//           window, from PIECES (the literal run or the match of a record, cut at 16-byte chunk borders)
//           in two lane-uniform rounds: round A, lane = chunk, writes the piece that covers the chunk's
//           first byte; round B, lane = piece, ORs in the head of every piece that starts inside a
//           chunk.  Literals are unaligned 16-byte reads from the compressed ring, matches from
//           the output ring (a match that overlaps itself reads an earlier period, far back, so long
//           runs do not serialise and never read a recycled slot).  Whether a source is final says one
//           byte per 16-byte chunk of the output ring: the tag of the ring lap whose bytes are final
//           there; a piece whose source is not is retried after the others.  Finished regions go to HBM
//           with one 16-byte store per lane (1 KB contiguous per wave).
//
// HBM/L2 traffic per block: without a table the compressed bytes are read by the pre-parse and once more
// by the mover, the record table is written and read once (16 B per sequence) and so are the token list
// and the region index; with a table the compressed bytes and the table (~3 % of the output) are read
// once, nothing is written but the output.  Matches and literals never touch HBM during the copy.
// No MFMA: byte moves.
#pragma once
#include "lz4_common.h"
#include "../lz4amd_params.h"
#include "lz4_preparse_kernel.h"

#ifdef LZ4AMD_TRACE
#define DTRACE(...) do { if (lane_here() == 0) { fprintf(stderr, "[w%u] ", wave_id()); fprintf(stderr, __VA_ARGS__); } } while (0)
#else
#define DTRACE(...) do {} while (0)
#endif

// developer build (-DLZ4AMD_DEC_TRACE, LZ4AMD_PROF=1 in the environment; tools/prof_trace.py): workgroup 0 logs what its copy waves do, when
#ifdef LZ4AMD_DEC_TRACE
// (stamps are kept in registers and written once, when the region is complete: a log write is a round trip to memory)
#define DSTAMP_DECL uint64_t rst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint32_t rsn_ = 0
#define DSTAMP(k_) do { if (!rst_[k_]) rst_[k_] = clock_ticks(); } while (0)
#define DSTAMP_LAST(k_) do { rst_[k_] = clock_ticks(); } while (0)
#define DCOUNT() do { rsn_++; } while (0)
#define DFLUSH(prof_, R_, np_) do { if ((prof_) && blockIdx.x == 0 && lane_here() == 0) { uint64_t* tr_ = (prof_) + (uint64_t)gridDim.x * 8; \
    const unsigned long long i_ = atomicAdd((unsigned long long*)tr_, 1ull); \
    if (i_ < (4u << 20) / 80 - 2) { uint64_t* q_ = tr_ + 1 + 10 * i_; q_[0] = (R_) | ((uint64_t)wave_id() << 32) | ((uint64_t)(np_) << 40) | ((uint64_t)rsn_ << 48); for (int j_ = 0; j_ < 8; j_++) q_[1 + j_] = rst_[j_]; } } } while (0)
#else
#define DSTAMP_DECL
#define DSTAMP(k_) do {} while (0)
#define DSTAMP_LAST(k_) do {} while (0)
#define DCOUNT() do {} while (0)
#define DFLUSH(prof_, R_, np_) do {} while (0)
#endif
#ifndef LZ4AMD_DEC_PROF
#ifdef LZ4AMD_DEC_TRACE
#define LZ4AMD_DEC_PROF 1
#else
#define LZ4AMD_DEC_PROF 0            // 1: the role profile of tools/prof_dec.py (LZ4AMD_PROF=1 in the environment) - a developer build, tools/build_variant.sh prof -DLZ4AMD_DEC_PROF=1
#endif
#endif
#ifndef LZ4AMD_DEC_CHUNK
#define LZ4AMD_DEC_CHUNK 16          // output bytes a lane composes at a time: 16 (regions of 1 KB) or 32 (regions of 2 KB; measured: DESIGN section 6)
#endif
#ifndef LZ4AMD_DEC_PARSE_GLOBAL
#define LZ4AMD_DEC_PARSE_GLOBAL 2    // 1: the parser waves read the stream from memory (L2), not from the compressed ring: how far they run ahead of the copy is then
                                     // not set by what the 32 KB ring holds beyond the regions in flight; 2: for large blocks that are mostly stream; 0: never
#endif
#ifndef LZ4AMD_DEC_COPYWAVES
#define LZ4AMD_DEC_COPYWAVES (LZ4AMD_DEC_MAXLEAD + 1 < 15 ? LZ4AMD_DEC_MAXLEAD + 1 : 15)      // no more than there can be regions in flight
#endif
#ifndef LZ4AMD_DEC_DMADEPTH
#define LZ4AMD_DEC_DMADEPTH 16
#endif
#ifndef LZ4AMD_DEC_CRKB
#define LZ4AMD_DEC_CRKB 32u           // compressed ring, KB (a power of two)
#endif
#ifndef LZ4AMD_DEC_MAXLEAD
#define LZ4AMD_DEC_MAXLEAD (LZ4AMD_DEC_CHUNK == 32 ? 10 : 15)        // regions in flight - 1 (what the LDS has room for: every region in flight is a slot of the output ring on top of the 64 KB window)
#endif
namespace lz4amd {

using DecBatch = ::lz4amd_dec_params;     // argument block (lz4amd_params.h)

using SeqRec = pre::SeqRec;               // { outpos, litpos, ll, off }, output positions biased by kBias

enum : uint32_t {
    kDecThreads = 1024,
    kDecWaves = kDecThreads / 64,
    kMoveWave = kDecWaves - 1,                  // compressed stream, sequence records, region index -> LDS
    kCopyWaves = kDecWaves - 1,                 // waves 0 .. kCopyWaves-1 may copy (fewer when the block comes with an entry-point table:
                                                //   kParsers waves below the mover then parse the stream from the table's entries, PARSER below);
                                                //   kActiveCopy of them do: no more than there can be regions in flight
    kChunk = LZ4AMD_DEC_CHUNK,                  // output bytes composed at a time by one lane (a region's fixed costs and a piece's address work are paid per
                                                //   lane-pass, not per byte: 16-byte chunks take 2 passes per KB of datagen -P60, 32-byte chunks ~1.25 - but with 2 KB
                                                //   regions more of the window is in flight, and more pieces find their source not final yet)
    kRegionShift = kChunk == 32 ? 11 : 10,
    kRegion = 1u << kRegionShift,               // 64 chunks
    kWindowSlots = 65536u >> kRegionShift,      // slots the 64 KB LZ4 window takes
    kMaxLead = LZ4AMD_DEC_MAXLEAD,              // a wave may lead the first unfinished region by this many
    kActiveCopy = LZ4AMD_DEC_COPYWAVES,         // copy waves that take regions (the others have nothing to do in stage B)
    kSlots = kWindowSlots + kMaxLead + 1,       // output ring slots (regions): the window + the regions in flight
    kRingBytes = kSlots * kRegion,
    kRingPad = 2 * kChunk,                      // mirror of the first bytes: reads (kChunk + 4 bytes from a dword boundary) never wrap
    kCrBytes = LZ4AMD_DEC_CRKB << 10,           // compressed ring (direct mapped: position mod 32 K)
    kCrPad = kChunk + 16,
    kRecCap = 2048,                             // sequence-record ring (1024 rows filled up on many-sequence data: HC-compressed 256 KiB blocks waited for room)
    kRecMask = kRecCap - 1,
    kIdxRing = kChunk == 32 ? 128 : 512,                             // first record of a region, per region (ring)
    kIdxMask = kIdxRing - 1,
    kEntRing = 256,                             // rows of the block's entry-point table (8 B each), ring; the mover brings 128 rows an instruction
    kEntMask = kEntRing - 1,
    kLaneSeqMax = 1024,                         // sequences between two rows of a table (lz4amd_k_compress writes a row every 2 to 16)
    kDmaDepth = LZ4AMD_DEC_DMADEPTH,            // LDS-DMA instructions (1 KB each) the mover keeps in flight
    kMaxTrips = kChunk == 32 ? 17 : 9,          // round-B trips per region (32 records each: a region of 2 KB starts at most 512 sequences + 2, one of 1 KB 258)
    kBias = pre::kBias,                         // output positions are biased: [kBias - prefix, kBias) is the history before dst
    kFirstRegion = kBias >> kRegionShift,
    kNone = 0xFFFFFFFFu,
};
static_assert(kChunk == 16 || kChunk == 32, "chunk");
static_assert((uint32_t)kRegionShift == (uint32_t)pre::kRegionShiftPre, "stage A's region index is per region of the copy stage");

// LDS carve-up of stage B (bytes); stage A uses the same memory before (lz4_preparse_kernel.h)
enum : uint32_t {
    kOffMisc = 0,                                            // u32[64] control words
    kOffMaskTab = kOffMisc + 64 * 4,                         // U32x8[33]: byte masks, entry n selects bytes [0, n) of a chunk
    kOffBits = kOffMaskTab + (kChunk + 1) * 32,                        // u8[kSlots * 64] per chunk of the output ring: the lap tag of the region whose bytes are final there
    kOffRegDone = kOffBits + kSlots * 64,                    // u32[kSlots] region + 1 that is complete in the slot
    kOffProg = kOffRegDone + kSlots * 4,                     // u32[kSlots] the slot's bell: counts up whenever the region in the slot has landed pieces
    kOffIdx = kOffProg + kSlots * 4,                         // u32[kIdxRing]
    kOffPend = (kOffIdx + kIdxRing * 4 + 7) & ~7u,           // per copy wave: u64[kMaxTrips] pending masks of round B, u32[16] pending pieces per chunk (a byte each)
    kPendStride = kMaxTrips * 8 + 64,
    kOffEnt = (kOffPend + kActiveCopy * kPendStride + 15) & ~15u,    // lz4amd_hint_entry[kEntRing]
    kOffRecs = kOffEnt + kEntRing * 8,                       // SeqRec[kRecCap]
    kOffCr = kOffRecs + kRecCap * 16,                        // compressed ring + pad
    kOffRing = kOffCr + kCrBytes + kCrPad,                   // output ring + pad
    kStreamLdsBytes = kOffRing + kRingBytes + kRingPad,
    kDecLdsBytes = kStreamLdsBytes > pre::kPreLdsBytes ? kStreamLdsBytes : pre::kPreLdsBytes,
};
static_assert(kDecLdsBytes <= 160u * 1024u, "LDS budget");
static_assert((kOffRecs % 16) == 0 && (kOffCr % 16) == 0 && (kOffRing % 16) == 0 && (kOffBits % 16) == 0 && (kOffPend % 8) == 0 && (kOffMaskTab % 32) == 0, "LDS alignment");

enum : uint32_t { M_BLOCK = 0, M_TWIN = 3, M_ABORT = 4, M_SPARE, M_CHI, M_CLO, M_HEAD, M_IHEAD, M_NEXT, M_OPEN,
                  M_EHEAD,       // rows of the entry-point table resident (mover -> parser)
                  M_PR0,         // first row the parser still needs (parser -> mover)
                  M_PBAD };      // the table does not fit the stream: the block is decoded again without it      // (M_HEAD, M_IHEAD: one aligned 64-bit word, published together: table rows resident, region index entries resident)      // (word 1 is the pre-parse's error word)

// scratch of one workgroup: the sequence-record table of the block it is decoding, the pre-parse's token list, the region index
__host__ __device__ inline uint64_t dec_scratch_bytes(uint32_t max_csize, uint32_t max_out) { return pre::scratch_bytes(max_csize, max_out); }

// ------------------------------------------------------------------------------ small helpers
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
// position in the compressed ring
__device__ __forceinline__ uint32_t mod_cr(uint32_t x) { return x & (kCrBytes - 1); }
// the compressed block's misalignment in memory
__device__ __forceinline__ uint32_t stream_misalign(lz4amd_gsrc src) { return (uint32_t)(uintptr_t)src & 15u; }
static_assert((kCrBytes & (kCrBytes - 1)) == 0, "ring size");
__device__ __forceinline__ uint32_t cr_fold(uint32_t a) { return umin32(a, a - kCrBytes); }        // [0, 2*32K) -> [0, 32K)
__device__ __forceinline__ uint32_t ring_fold(uint32_t a) { return umin32(a, a - kRingBytes); }
// a control word every lane of the wave agrees on
__device__ __forceinline__ uint32_t uload(const uint32_t* w) { return __builtin_amdgcn_readfirstlane(lds_load_acquire(w)); }

using pre::chunk_byte;
using pre::chunk_set_byte;
using pre::load_granule;
// A chunk in registers: two register quads (the second one only with 32-byte chunks)
struct U32x8 { U32x4 a, b; };
constexpr bool kWide = kChunk == 32;
__device__ __forceinline__ U32x8 zero8() { U32x8 z; z.a[0] = z.a[1] = z.a[2] = z.a[3] = 0; z.b = z.a; return z; }
// dword k (0..7) of the mask that selects bytes [0, n) of a chunk, n in 0..kChunk
__device__ __forceinline__ uint32_t low_bytes_mask(uint32_t n, uint32_t k) {
    const int32_t r = (int32_t)n - 4 * (int32_t)k;
    return r >= 4 ? 0xFFFFFFFFu : (r <= 0 ? 0u : ((1u << (8 * r)) - 1u));
}
// row n of the LDS mask table (32 bytes a row): bytes [0, n) of a chunk (computing the dword masks took ~60 instructions per call, three
// calls per region - a fifth of the copy stage's instruction count; the rows are read before the piece's source, so that both travel together)
__device__ __forceinline__ U32x8 mask_row(const char* smem, uint32_t n) {
    const U32x4* p = (const U32x4*)(smem + kOffMaskTab + 32 * n);
    U32x8 r; r.a = p[0]; if constexpr (kWide) r.b = p[1]; else r.b = r.a; return r;
}
// v restricted to bytes [lo, hi): mh = row hi, ml = row lo
__device__ __forceinline__ U32x8 and_rows(const U32x8& v, const U32x8& mh, const U32x8& ml) {
    U32x8 r;
    r.a[0] = v.a[0] & mh.a[0] & ~ml.a[0]; r.a[1] = v.a[1] & mh.a[1] & ~ml.a[1]; r.a[2] = v.a[2] & mh.a[2] & ~ml.a[2]; r.a[3] = v.a[3] & mh.a[3] & ~ml.a[3];
    if constexpr (kWide) { r.b[0] = v.b[0] & mh.b[0] & ~ml.b[0]; r.b[1] = v.b[1] & mh.b[1] & ~ml.b[1]; r.b[2] = v.b[2] & mh.b[2] & ~ml.b[2]; r.b[3] = v.b[3] & mh.b[3] & ~ml.b[3]; }
    else r.b = r.a;
    return r;
}
__device__ __forceinline__ U32x8 and_row(const U32x8& v, const U32x8& mh) {
    U32x8 r;
    r.a[0] = v.a[0] & mh.a[0]; r.a[1] = v.a[1] & mh.a[1]; r.a[2] = v.a[2] & mh.a[2]; r.a[3] = v.a[3] & mh.a[3];
    if constexpr (kWide) { r.b[0] = v.b[0] & mh.b[0]; r.b[1] = v.b[1] & mh.b[1]; r.b[2] = v.b[2] & mh.b[2]; r.b[3] = v.b[3] & mh.b[3]; }
    else r.b = r.a;
    return r;
}
__device__ __forceinline__ U32x8 keep_bytes(const char* smem, const U32x8& v, uint32_t lo, uint32_t hi) { return and_rows(v, mask_row(smem, hi), mask_row(smem, lo)); }
__device__ __forceinline__ U32x8 keep_low_bytes(const char* smem, const U32x8& v, uint32_t hi) { return and_row(v, mask_row(smem, hi)); }
// kChunk bytes starting at ANY byte a of an LDS array of dwords (the arrays are padded: a + kChunk + 4 is in range): aligned dwords
// and byte alignments - an LDS access that is not aligned to its own width is served lane by lane on gfx950
__device__ __forceinline__ U32x8 lds_read_chunk_at(const uint8_t* base, uint32_t a) {
    const uint32_t* r32 = (const uint32_t*)(base + (a & ~3u));
    const uint32_t sh = a & 3u;
    const uint32_t d0 = r32[0], d1 = r32[1], d2 = r32[2], d3 = r32[3], d4 = r32[4];
    U32x8 v;
    v.a[0] = align_bytes(d1, d0, sh); v.a[1] = align_bytes(d2, d1, sh); v.a[2] = align_bytes(d3, d2, sh); v.a[3] = align_bytes(d4, d3, sh);
    if constexpr (kWide) {
        const uint32_t d5 = r32[5], d6 = r32[6], d7 = r32[7], d8 = r32[8];
        v.b[0] = align_bytes(d5, d4, sh); v.b[1] = align_bytes(d6, d5, sh); v.b[2] = align_bytes(d7, d6, sh); v.b[3] = align_bytes(d8, d7, sh);
    } else v.b = v.a;
    return v;
}

// The control words M_ABORT .. M_OPEN (8 consecutive dwords) in one look: two 16-byte LDS reads issued together, so a
// wave pays one LDS round trip instead of one per word.
struct Ctl { uint32_t abort_, chi, ihead, head, clo, next, open; };
__device__ __forceinline__ Ctl ctl_unpack(const U32x4& a, const U32x4& b) {
    Ctl c;
    c.abort_ = __builtin_amdgcn_readfirstlane(a[0]);
    c.chi = __builtin_amdgcn_readfirstlane(a[2]); c.clo = __builtin_amdgcn_readfirstlane(a[3]);
    c.head = __builtin_amdgcn_readfirstlane(b[0]); c.ihead = __builtin_amdgcn_readfirstlane(b[1]);
    c.next = __builtin_amdgcn_readfirstlane(b[2]); c.open = __builtin_amdgcn_readfirstlane(b[3]);
    return c;
}
__device__ __forceinline__ Ctl ctl_snapshot(const char* smem) {
    const U32x4* w = (const U32x4*)(smem + kOffMisc + 4 * M_ABORT);
    U32x4 a, b;
    lds_load_pair16(w, a, b);
    return ctl_unpack(a, b);
}
// ... and in the same look the region index entries of regions R and R + 1 (read AFTER the control words: they are
// valid if the control words say so)
__device__ __forceinline__ Ctl ctl_snapshot_region(const char* smem, uint32_t R, uint32_t& i0, uint32_t& i1) {
    const U32x4* w = (const U32x4*)(smem + kOffMisc + 4 * M_ABORT);
    const uint32_t* idx = (const uint32_t*)(smem + kOffIdx);
    U32x4 a, b;
    lds_load_pair16_then2(w, a, b, &idx[R & kIdxMask], &idx[(R + 1) & kIdxMask], i0, i1);
    i0 = __builtin_amdgcn_readfirstlane(i0); i1 = __builtin_amdgcn_readfirstlane(i1);
    return ctl_unpack(a, b);
}
static_assert(M_OPEN == M_ABORT + 7 && (kOffMisc + 4 * M_ABORT) % 16 == 0, "control word layout");

// First region that is not complete: every region below it is final.  (Kept by the copy waves: whoever completes a
// region moves the word over every complete region in front of it.)
__device__ __forceinline__ uint32_t first_open_region(const char* smem) {
    return uload((const uint32_t*)(smem + kOffMisc) + M_OPEN);
}

// ------------------------------------------------------------------------------ MOVER
// One wave moves everything the copy reads into LDS, as far ahead as the rings allow, and does nothing else:
//   * the compressed block -> the 32 KB ring the literals are read from,
//   * the record table -> the 2048-row record ring,
//   * the pre-parse's region index (first record of every 1 KB region of output) -> its 512-entry ring, 64 a trip.
// Stream and records travel by LDS-DMA (global_load_lds_dwordx4: 1 KB per instruction, no registers in between), up to
// kDmaDepth KB in flight; the wave keeps the order of what it issued in a bit queue and publishes what has landed.
// The stream is fetched in the ALIGNED 16-byte granules of its memory (position P sits at ring address P + mis, mis =
// the block's misalignment): the first and the last granule reach up to 15 bytes beyond the block, inside the same
// granule (and page); those bytes are never used.  No per-record work: whatever the copy needs to know about a region it
// reads from the rings.  Nothing here blocks on the copy: what does not fit now is tried on the next trip.
// With an entry-point table (hint != null) records and region index are made by the PARSER wave, in LDS: the mover then
// moves the table's rows (a 256-row ring, 64 rows per DMA instruction) instead of records and index.
__device__ __forceinline__ void mover_role(lz4amd_gsrc src, uint32_t csize, const SeqRec* rectab, const uint32_t* ridx, uint32_t nseq, uint32_t rend, char* smem,
                                           lz4amd_gsrc hint, uint32_t nent) {
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    uint32_t* idx = (uint32_t*)(smem + kOffIdx);
    const SeqRec* recs = (const SeqRec*)(smem + kOffRecs);
    const uint32_t* regdone = (const uint32_t*)(smem + kOffRegDone);
    char* cr = smem + kOffCr;
    const uint32_t lane = lane_here();
    wave_priority_high();                              // fifteen waves wait for what this one produces
    const uint32_t nrows = nseq + 1;                   // (the table ends with a sentinel row: where the block's output ends)
    const uint32_t mis = stream_misalign(src);
    lz4amd_gsrc src0 = src - mis;                      // 16-byte aligned
    const uint32_t ngran = (mis + csize + 15) >> 4, nchunks = (ngran + 63) >> 6;      // granules, 1 KB chunks of the stream
    uint32_t si = 0, sc = 0;                           // stream chunks issued / landed
    uint32_t hi = 0, hc = 0;                           // table rows issued / landed (multiples of 64)
    uint32_t ihead = kFirstRegion, tail = 0, need = 0;
    uint64_t fifo = 0; uint32_t q = 0;                 // what is in flight, oldest at bit 0: 0 = a stream chunk, 1 = 64 table rows
    uint32_t pidx = 0, icarry = 0;
    const bool hinted = hint != nullptr;
    char* ent = smem + kOffEnt;
    const uint32_t crBytes = kCrBytes;
    uint32_t ei = 0, ec = 0;                           // table rows issued / landed (multiples of 64)
    if (!hinted && ihead + lane < rend) pidx = ridx[lane];
    for (;;) {
        const Ctl c = ctl_snapshot(smem);
        if (c.abort_) { vmem_wait<0>(); break; }
        bool progress = false;
        const uint32_t pr0 = hinted ? uload(&misc[M_PR0]) : 0u;
        if (hinted) { ihead = c.ihead; hc = c.head; }  // (published by the parser)
        // ---- the first open region: this wave is the only one that moves the word, over every complete region in front of it
        //      (at most kMaxLead + 1 regions are in flight; their complete marks in one read, lane = region)
        uint32_t g = c.open;
        {
            const uint32_t gl = g + lane;
            const bool done_l = lane <= kMaxLead && gl < rend && lds_load_acquire(&regdone[gl % kSlots]) == gl + 1;
            const unsigned long long m = __ballot(done_l);
            const uint32_t adv = (uint32_t)__ffsll((long long)~m) - 1;          // complete regions at the front
            if (adv) { g += adv; if (lane == 0) lds_store_release(&misc[M_OPEN], g); progress = true; }
        }
        // ---- the region index: regions [ihead, ihead + 64), once the ring has let go of the regions 512 below them
        if (!hinted && ihead < rend) {
            const uint32_t n = rend - ihead < 64 ? rend - ihead : 64;
            if (ihead + n <= g + kIdxRing) {
                // (a sequence that covers several region starts noted only the first: holes take the entry before them)
                const uint32_t filled = wave_incl_max_u32(pidx > icarry ? pidx : icarry);
                icarry = wave_readlane(filled, 63);
                if (lane < n) idx[(ihead + lane) & kIdxMask] = filled - 1;
                ihead += n;
                pidx = 0;
                if (ihead + lane < rend) pidx = ridx[ihead - kFirstRegion + lane];
                progress = true;
            }
        }
        wave_lds_fence();
        // ---- what the copy still needs: the first record and the first literal byte of the first open region g (both only
        //      ever move forward, so stale values are merely careful: less room)
        if (g >= rend) need = csize;
        else if (g < ihead) {
            tail = __builtin_amdgcn_readfirstlane(idx[g & kIdxMask]);
            if (tail < hc) {
                const SeqRec r = recs[tail & kRecMask];
                uint32_t d = (g << kRegionShift) - r.outpos; if (d > r.ll) d = r.ll;
                need = __builtin_amdgcn_readfirstlane(r.litpos + d);
            }
        } else if (hinted && c.clo > need) need = c.clo;         // every region that has records is complete: what is needed next is the token the parser stands at
        // ---- issue: stream chunk si may be written once nobody needs the bytes 32 K below its last one; rows [hi, hi + 64)
        //      once the ring has let go of the rows 2048 below them
        bool issued = false;
        while (q < kDmaDepth) {
            bool did = false;
            if (si < nchunks && ((si + 1) << 10) <= need + mis + crBytes) {
                uint32_t gr = 64 * si + lane; if (gr >= ngran) gr = ngran - 1;
                lds_dma16((const void*)(src0 + 16 * (uint64_t)gr), cr + ((si << 10) & (crBytes - 1)));
                q++; si++; did = true;                                               // (its fifo bit is 0)
            }
            if (!hinted && q < kDmaDepth && hi < nrows && hi + 64 - tail <= kRecCap) {
                lds_dma16(rectab + (hi + lane < nrows ? hi + lane : nrows - 1), (char*)recs + ((hi & kRecMask) << 4));      // (rows behind the table's last: the last again, nobody reads them)
                fifo |= 1ull << q; q++; hi += 64; did = true;
            }
            if (hinted && q < kDmaDepth && ei < nent && ei + 128 <= pr0 + kEntRing) {
                // (a lane brings two rows: 16 bytes; behind the table's last pair of rows the last pair again - the table's memory is a multiple of 16 bytes)
                const uint32_t pair = (ei >> 1) + lane, lastp = (nent - 1) >> 1;
                lds_dma16((const void*)(hint + 16 * (uint64_t)(pair < lastp ? pair : lastp)), ent + ((ei & kEntMask) << 3));
                fifo |= 1ull << q; q++; ei += 128; did = true;
            }
            if (!did) break;
            issued = true;
        }
        if (issued) progress = true;                         // (what was just issued is looked after on the next trip, without a pause)
        // ---- retire: busy = all but the youngest half of the queue, idle = everything
        uint32_t keep = q;
        if (!issued) { vmem_wait<0>(); keep = 0; }
        else if (q > kDmaDepth / 2) { vmem_wait<kDmaDepth / 2>(); keep = kDmaDepth / 2; }
        if (keep < q) {
            const uint32_t r = q - keep;
            const uint32_t nrec = (uint32_t)__popcll(fifo & ((1ull << r) - 1ull));
            const uint32_t s1 = sc + (r - nrec);
            // the ring's pad mirrors its first 32 bytes: reads never wrap
            if (((sc + (crBytes >> 10) - 1) & ~((crBytes >> 10) - 1)) < s1) {            // (a chunk that starts a lap has landed)
                if (lane < kCrPad / 16) *(U32x4*)(cr + crBytes + 16 * lane) = *(const U32x4*)(cr + 16 * lane);
            }
            sc = s1; if (hinted) ec += 128 * nrec; else hc += 64 * nrec;
            fifo >>= r; q = keep;
            wave_lds_fence();
            if (lane == 0) {
                const uint32_t got = sc << 10;
                lds_store_release(&misc[M_CHI], got <= mis ? 0u : (got - mis < csize ? got - mis : csize));
            }
            progress = true;
        }
        if (progress) {
            wave_lds_fence();
            if (hinted) { if (lane == 0) lds_store_release(&misc[M_EHEAD], ec < nent ? ec : nent); }
            else if (lane == 0) lds_store_release64((uint64_t*)&misc[M_HEAD], (uint64_t)(hc < nrows ? hc : nrows) | ((uint64_t)ihead << 32));
        }
        if (hinted ? (ec >= nent && sc == nchunks && g >= rend) : (hc >= nrows && ihead == rend && sc == nchunks && g >= rend)) {          // everything moved, every region complete
            vmem_wait<0>();                                  // (nothing of this block may land in LDS later)
            break;
        }
        if (!progress) { if (hinted) spin_pause(); else { spin_pause_long(); if (sc < nchunks || hc < nrows) spin_pause_long(); } }      // (the rings hold tens of thousands of cycles of work: a look every ~1000 is plenty, and every look takes issue slots from the copy waves of this SIMD)
    }
}

// ------------------------------------------------------------------------------ PARSER (blocks that come with an entry-point table)
// An entry-point table (lz4amd_params.h: lz4amd_hint_entry; written by lz4amd_k_compress next to the block it made, or by
// anybody else) names one sequence of the token chain every few sequences (lz4amd_k_compress: about every 512 bytes of output).  With it the serial chain is cut in pieces that
// are parsed side by side: LANE k of this wave walks the sequences from row r0 + k up to row r0 + k + 1 out of the
// compressed ring in LDS - token, literal length, offset, match length, the same rules as the pre-parse's P5
// (read_variable_length lz4.c:1979-2014; lz4.c:2279, 2312-2318, 2356, 2423) - and writes their records and the region
// index straight into the rings the copy waves read.  Stage A, its scratch in HBM and its two passes over the stream
// are not needed at all.
// The table is NEVER trusted: row 0 must be the block's first byte, every lane must arrive exactly at the next row (token
// position, output position and sequence count), the last row must be the block's end - by induction every record then
// lies on the true chain - and nothing a batch of lanes wrote is published before all of them have arrived.  A table
// that does not fit the stream costs time only: M_PBAD, and the block is decoded again the ordinary way.
// Bytes that are not in the ring (a row far ahead of what the mover has loaded; length fields of many bytes) are read
// from memory instead.
struct HintEnt { uint32_t tok, out, ord; };      // a row as the walk uses it: ord (sequences before it) is counted up from the rows' 8-bit differences

// four stream bytes from position x on, out of the compressed ring (two aligned dwords and a byte alignment; the ring's pad
// covers the dword behind its end)
__device__ __forceinline__ uint32_t cr_fetch4(const char* cr, uint32_t x, uint32_t mis) {
    const uint32_t a = mod_cr(x + mis);
    const uint32_t* d = (const uint32_t*)(cr + (a & ~3u));
    return align_bytes(d[1], d[0], a & 3u);
}
// eight stream bytes from position x on (three aligned dwords and two byte alignments; the pad covers the dwords behind the ring's end)
__device__ __forceinline__ uint64_t cr_fetch8(const char* cr, uint32_t x, uint32_t mis) {
    const uint32_t a = mod_cr(x + mis);
    const uint32_t* d = (const uint32_t*)(cr + (a & ~3u));
    const uint32_t d0 = d[0], d1 = d[1], d2 = d[2];
    return (uint64_t)align_bytes(d1, d0, a & 3u) | ((uint64_t)align_bytes(d2, d1, a & 3u) << 32);
}
// ... the same from memory (LZ4AMD_DEC_PARSE_GLOBAL): never a byte outside src[0, csize) - a look that would run over the block's end is
// moved back to end there (its value is then not the stream at x: every caller that can get there takes the careful form, which does not use it)
__device__ __forceinline__ uint32_t g_fetch4(lz4amd_gsrc src, uint32_t x, uint32_t csize) {
    if (csize < 4) return 0;
    uint32_t v; __builtin_memcpy(&v, src + (x + 4 <= csize ? x : csize - 4), 4); return v;
}
__device__ __forceinline__ uint64_t g_fetch8(lz4amd_gsrc src, uint32_t x, uint32_t csize) {
    if (csize < 8) return 0;
    uint64_t v; __builtin_memcpy(&v, src + (x + 8 <= csize ? x : csize - 8), 8); return v;
}
// one stream byte at position x (x < csize): out of the compressed ring when it is resident, else from memory
__device__ __forceinline__ uint32_t pbyte(const char* cr, lz4amd_gsrc src, uint32_t x, uint32_t mis, uint32_t chi) {
    if (x < chi) return (uint32_t)*(const uint8_t*)(cr + mod_cr(x + mis));
    return (uint32_t)src[x];
}
// A length field's extension bytes (lz4.c:1979-2014): acc += bytes from pos on, up to and including the first one that is
// not 255.  A byte at position x may be read iff x + tailroom <= csize (16 for literal lengths, 5 for match lengths: what
// the reference's ilimit / iend - 4 tests say); else the block is malformed.  The first bytes lane by lane (nearly all
// fields end there), the rest of a long field 64 bytes at a time by the whole wave, from memory.
__device__ __forceinline__ void ext_field(bool need, uint32_t& pos, uint32_t& acc, bool& bad, uint32_t tailroom,
                                          const char* cr, lz4amd_gsrc src, uint32_t csize, uint32_t mis, uint32_t chi) {
    const uint32_t lane = lane_here();
    bool more = need;
    for (uint32_t trip = 0; trip < 4 && __any(more); trip++) {
        const bool legal = pos + tailroom <= csize;
        uint32_t x = 0;
        if (more && legal) x = pbyte(cr, src, pos, mis, chi);
        bad = bad || (more && !legal);
        acc += more && legal ? x : 0u; pos += more && legal ? 1u : 0u;
        more = more && legal && x == 255;
    }
    unsigned long long pm = __ballot(more);
    while (pm) {
        const uint32_t l = (uint32_t)__ffsll((long long)pm) - 1; pm &= pm - 1;
        uint32_t P0 = wave_readlane(pos, l), add = 0; bool fbad = false;
        for (;;) {
            const uint32_t ps = P0 + lane;
            const bool legal = ps + tailroom <= csize;
            const uint32_t x = legal ? (uint32_t)src[ps] : 0u;
            const unsigned long long stopm = __ballot(!legal || x != 255);
            if (!stopm) { add += 255 * 64; P0 += 64; if (add > 0x7F000000u) { fbad = true; break; } continue; }
            const uint32_t kk = (uint32_t)__ffsll((long long)stopm) - 1;
            if (!wave_readlane(legal ? 1u : 0u, kk)) fbad = true;
            else { add += 255 * kk + wave_readlane(x, kk); P0 += kk + 1; }
            break;
        }
        if (lane == l) { acc += add; pos = P0; bad = bad || fbad; }
    }
}

// One step of a lane's walk: the sequence whose token is at W.p, by the decoder's rules.  The FAST form covers what nearly
// every sequence is - a literal length of at most seven extension bytes (below 1800: poorly compressible data is made of
// literal runs of hundreds of bytes, and ONE lane that needs the careful form sends the whole wave through it), a match
// length of at most two (below 529), not in the block's last 24 bytes, every byte in the ring: the token and the seven bytes
// behind it in one look (three aligned dwords), the offset and the two bytes behind it in another.  Whatever it does not cover takes the CAREFUL form: the same
// rules, any field length, bytes from memory when they are not resident, the block's last sequence.
struct Walk { uint32_t p, o, i, end, oend, iend; uint64_t W; bool act, res, bad; };      // res: every byte of the lane's row is in the ring; W: the eight stream bytes at p (asked for as soon as p is known: the walk is a chain of dependent LDS round trips)
struct StepOut { SeqRec r; uint32_t oe; bool ok; };
struct PCtx { const char* cr; lz4amd_gsrc src; uint32_t csize, mis, chi, capB, low; bool pg; };      // pg: the walk reads the stream from memory, not from the compressed ring
__device__ __forceinline__ StepOut parser_step(const PCtx& X, Walk& w, uint32_t& n_careful) {
    StepOut S; S.r.outpos = w.o; S.r.litpos = 0; S.r.ll = 0; S.r.off = 0; S.oe = w.o; S.ok = false;
    uint32_t pn = w.p;
    bool careful = w.act && !w.res;
    {
        const uint32_t W = (uint32_t)w.W;
        const uint32_t b = W & 0xFFu, e1 = (W >> 8) & 0xFFu, e2 = (W >> 16) & 0xFFu, e3 = W >> 24;
        const bool lx = (b >> 4) == 15, lx2 = lx && e1 == 255, lx3 = lx2 && e2 == 255;
        uint32_t ll = (b >> 4) + (lx ? e1 : 0u) + (lx2 ? e2 : 0u) + (lx3 ? e3 : 0u);
        uint32_t q = w.p + 1 + (lx ? 1u : 0u) + (lx2 ? 1u : 0u) + (lx3 ? 1u : 0u);
        bool lmore = lx3 && e3 == 255;                                           // the field goes on behind the token's dword
        if (__any(w.act && lmore)) {
            // ... into the four bytes behind: k of them 255, then the one that ends the field (none of the four: the careful form)
            const uint32_t W1 = (uint32_t)(w.W >> 32), nz = ~W1;
            const uint32_t k = nz ? ((uint32_t)__ffs((int)nz) - 1u) >> 3 : 4u;
            const uint32_t ek = k < 4 ? (W1 >> (8u * k)) & 0xFFu : 0u;
            ll += lmore ? 255u * k + ek : 0u;
            q += lmore ? (k < 4 ? k + 1u : 4u) : 0u;
            lmore = lmore && k == 4;
        }
        const uint32_t m = q + ll;
        // (m + 24 <= csize: not the last sequence, lz4.c:2279, and every length byte looked at may be read, lz4.c:1986-2006)
        careful = careful || (w.act && (lmore || m + 24 > X.csize || X.capB - w.o < ll + kMfLimit));
        const uint32_t V = X.pg ? g_fetch4(X.src, m, X.csize) : cr_fetch4(X.cr, m, X.mis);
        const uint32_t off = V & 0xFFFFu, f1 = (V >> 16) & 0xFFu, f2 = V >> 24;
        const bool mx = (b & 15u) == 15, mx2 = mx && f1 == 255;
        const uint32_t ml = (b & 15u) + (mx ? f1 : 0u) + (mx2 ? f2 : 0u) + kMinMatch;
        careful = careful || (w.act && mx2 && f2 == 255);
        const bool fast = w.act && !careful;
        const uint32_t ms = w.o + ll;
        const bool fbad = off - 1u >= ms - X.low || X.capB - ms < ml + kLastLiterals || w.i >= w.iend;      // lz4.c:2356 (offset 0 wraps), 2423; more sequences than the rows say
        w.bad = w.bad || (fast && fbad);
        S.ok = fast && !fbad;
        S.r.litpos = q; S.r.ll = ll; S.r.off = off;
        S.oe = S.ok ? ms + ml : S.oe;
        pn = S.ok ? m + 2 + (mx ? 1u : 0u) + (mx2 ? 1u : 0u) : pn;
        w.W = X.pg ? g_fetch8(X.src, pn, X.csize) : cr_fetch8(X.cr, pn, X.mis);                  // the next token, on its way while this sequence's record is written
    }
    if (__any(careful)) {
        n_careful++;
        const bool on = careful;
        uint32_t b = 0;
        if (on) b = pbyte(X.cr, X.src, w.p, X.mis, X.chi);
        uint32_t ll = b >> 4, q = w.p + 1;
        bool cbad = false;
        ext_field(on && ll == 15, q, ll, cbad, 16, X.cr, X.src, X.csize, X.mis, X.chi);
        cbad = cbad || (on && ll > X.csize);
        bool go = on && !cbad;
        const uint32_t rem = X.csize - q, room = X.capB - w.o;
        const bool last = rem < ll + 8 || room < ll + kMfLimit;                 // lz4.c:2279
        cbad = cbad || (go && last && !(rem == ll && room >= ll));               // lz4.c:2312-2318
        const bool mt = go && !cbad && !last;
        const uint32_t m = q + ll;                                              // (mt: m + 8 <= csize)
        uint32_t off = 0;
        if (mt) off = pbyte(X.cr, X.src, m, X.mis, X.chi) | (pbyte(X.cr, X.src, m + 1, X.mis, X.chi) << 8);
        uint32_t ml = b & 15, nx = m + 2;
        ext_field(mt && ml == 15, nx, ml, cbad, 5, X.cr, X.src, X.csize, X.mis, X.chi);
        ml += kMinMatch;
        const uint32_t ms = w.o + ll;
        cbad = cbad || (mt && (off == 0 || off > ms - X.low || X.capB - ms < ml + kLastLiterals));      // lz4.c:2356, 2423
        cbad = cbad || (on && w.i >= w.iend);
        go = on && !cbad;
        w.bad = w.bad || (on && cbad);
        if (go) { S.r.litpos = q; S.r.ll = ll; S.r.off = last ? 0u : off; S.oe = last ? ms : ms + ml; pn = last ? X.csize : nx; S.ok = true; }
        if (on) w.W = X.pg ? g_fetch8(X.src, pn, X.csize) : cr_fetch8(X.cr, pn, X.mis);
    }
    w.p = pn;
    return S;
}

// kParsers waves parse side by side.  A wave CLAIMS the next batch of rows under a lock (as many rows as are resident and
// fit the rings, one per lane), walks it, and PUBLISHES it when every earlier batch is published (a ticket per claim):
// publication in order is what makes the induction work - a batch's first row is true because the batch before arrived
// at it.  A wave's walk is a chain of dependent steps of ~1000 cycles each (its SIMD's VALU is kept busy by the copy waves
// it shares it with: an instruction of the lone walker waits for theirs).  Measured on 256 x 4 MiB with the compressor's
// tables (rows of at most 8 sequences): two parser waves 1.18 / 1.47 / 1.54 ms per GiB at P60 / P90 / P20, three 1.22 /
// 1.54 / 1.50, four 1.24 / 1.61 / 1.55: the decoder is bound by VALU issue, and a third walker's instructions cost the
// twelve copy waves more than its rows bring.
#ifndef LZ4AMD_PARSERS
#define LZ4AMD_PARSERS 2
#endif
enum : uint32_t { kParsers = LZ4AMD_PARSERS, kFirstParseWave = kDecWaves - 1 - kParsers };
enum : uint32_t { P_NEXT = 16, P_RZ, P_TICKET, P_TURN, P_LOCK, P_ICARRY, P_ORD };      // (P_ORD: sequences before row P_NEXT)      // shared words of the parser waves (misc[]); the first four are
static_assert(M_EHEAD == M_ABORT + 8 && P_NEXT == M_ABORT + 12, "the words a claim looks at are 64 consecutive bytes");      // read with M_ABORT .. M_PBAD in one trip to the LDS
__device__ __forceinline__ void parser_unlock(uint32_t* misc) { wave_lds_fence(); if (lane_here() == 0) lds_store_release(&misc[P_LOCK], 0u); }
__device__ __forceinline__ void parser_role(lz4amd_gsrc src, uint32_t csize, uint32_t cap, uint32_t prefix, uint32_t total, uint32_t nseq,
                                            uint32_t nreg, uint32_t rend, char* smem, uint64_t* prof) {
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    uint32_t* idx = (uint32_t*)(smem + kOffIdx);
    SeqRec* recs = (SeqRec*)(smem + kOffRecs);
    const lz4amd_hint_entry* ent = (const lz4amd_hint_entry*)(smem + kOffEnt);
    const uint32_t lane = lane_here();
    PCtx X; X.cr = smem + kOffCr; X.src = src; X.csize = csize; X.mis = stream_misalign(src); X.chi = 0; X.capB = cap + kBias; X.low = kBias - prefix;
    // (a block that is mostly stream - datagen -P60 and less compressible - and large: how far the walk may run ahead of the copy is else set by what
    //  the 32 KB ring holds beyond the regions in flight; blocks of long matches walk faster out of the LDS)
    X.pg = LZ4AMD_DEC_PARSE_GLOBAL == 2 ? (total >= (1u << 20) && csize > (total >> 1) - (total >> 4)) : LZ4AMD_DEC_PARSE_GLOBAL != 0;
    wave_priority_high();                              // the copy waves wait for what these waves produce
    uint32_t tail = 0, stall = 0;
    bool fail = false;
    uint32_t n_batch = 0, n_lanes = 0, n_steps = 0, n_careful = 0; uint64_t t_wait = 0, t_walk = 0, tq = prof ? clock_ticks() : 0;      // developer profile
#ifdef LZ4AMD_PROF_PARSER
    uint64_t t_lock = 0, t_sel = 0, t_turn = 0, t_pub = 0, tp = clock_ticks(); uint32_t n_nowork = 0;
#define PSTAMP(acc) do { const uint64_t t_ = clock_ticks(); acc += t_ - tp; tp = t_; } while (0)
#else
#define PSTAMP(acc) do {} while (0)
#endif
    for (;;) {
        // ---- claim the next batch of rows (one wave at a time)
        PSTAMP(t_pub);
        for (;;) {
            uint32_t got = 0;
            if (lane == 0) got = atomicCAS(&misc[P_LOCK], 0u, 1u) == 0u ? 1u : 0u;
            if (__builtin_amdgcn_readfirstlane(got)) break;
            if (uload(&misc[M_ABORT])) return;
            spin_pause();
        }
        wave_lds_fence();
        PSTAMP(t_lock);
        // (everything a claim looks at - M_ABORT .. M_PBAD and the parsers' own four words - in ONE trip to the LDS: the claim is a
        //  chain of dependent round trips of a wave whose every instruction waits for the copy waves', and it is made under the lock)
        U32x4 qa, qb, qc, qd;
        lds_load_quad16((const U32x4*)(smem + kOffMisc + 4 * M_ABORT), qa, qb, qc, qd);
        const Ctl c = ctl_unpack(qa, qb);
        const uint32_t r0 = __builtin_amdgcn_readfirstlane(qd[0]);
        if (c.abort_ || r0 >= nreg) { parser_unlock(misc); break; }
        const uint32_t Ra = __builtin_amdgcn_readfirstlane(qd[1]);            // regions below Ra belong to the batches claimed before
        const uint32_t ticket = __builtin_amdgcn_readfirstlane(qd[2]), turn = __builtin_amdgcn_readfirstlane(qd[3]);
        const uint32_t ehead = __builtin_amdgcn_readfirstlane(qc[0]);
        const uint32_t g = c.open;
        X.chi = X.pg ? 0u : c.chi;            // (0: every byte the careful form looks at comes from memory)
        // how many lanes?  Rows [r0, r0 + nl] must be resident; the lanes' records must fit the record ring behind the first
        // record an open region still needs, their regions the index ring; and the lanes' stream bytes should be resident.
        uint32_t nlmax = nreg - r0 < 64 ? nreg - r0 : 64;
        const uint32_t er = ehead > r0 + 1 ? ehead - r0 - 1 : 0;
        if (er < nlmax) nlmax = er;
        HintEnt A, B; A.tok = A.out = A.ord = 0; B = A;
        uint32_t dseq = 0;                              // sequences between my row and the next
        if (lane < nlmax) {
            const lz4amd_hint_entry a = ent[(r0 + lane) & kEntMask], b = ent[(r0 + lane + 1) & kEntMask];
            A.tok = a.tok_ord & 0xFFFFFFu; A.out = a.out; B.tok = b.tok_ord & 0xFFFFFFu; B.out = b.out;
            dseq = ((b.tok_ord >> 24) - (a.tok_ord >> 24)) & 0xFFu;
            A.ord = a.tok_ord >> 24;                    // (for now: its low 8 bits, checked against the count below)
        }
        const uint32_t ord0 = uload(&misc[P_ORD]);      // sequences before row r0 (counted by the batches claimed before; under the lock)
        const uint32_t incl = wave_incl_sum_u32(dseq);
        const bool lowbad = lane < nlmax && A.ord != ((ord0 + incl - dseq) & 0xFFu);      // (the row's own 8 bits must be what the count says)
        A.ord = ord0 + incl - dseq; B.ord = A.ord + dseq;
        const uint32_t tail_now = idx[g & kIdxMask];       // (asked for with the rows: one trip; it counts only if the entry is there)
        // the rows themselves: never decreasing, inside the block, no more sequences between two of them than 512 bytes of output can start
        bool rowbad = lane < nlmax && (A.tok > B.tok || A.out > B.out || A.ord > B.ord || B.tok > csize || B.out > total || B.ord - A.ord > kLaneSeqMax
                                       || (A.tok == B.tok) != (A.ord == B.ord) || (A.tok == B.tok && A.out != B.out));
        if (r0 == 0 && lane == 0 && nlmax && (A.tok | A.out | A.ord)) rowbad = true;
        rowbad = rowbad || lowbad;
        if (r0 + lane + 1 == nreg && lane < nlmax && (B.tok != csize || B.out != total || B.ord != nseq)) rowbad = true;
        if (__any(rowbad)) { fail = true; parser_unlock(misc); break; }
        if (g < c.ihead) tail = __builtin_amdgcn_readfirstlane(tail_now);               // first record an open region needs (only ever grows)
        const uint32_t Rb = (B.out + kBias + kRegion - 1) >> kRegionShift;              // regions below Rb have their first byte before my lane's end
        const bool okrec = B.ord + 1 - tail <= kRecCap;
        const bool okidx = Rb + 1 <= g + kIdxRing;
        const unsigned long long mh = __ballot(lane < nlmax && okrec && okidx), mr = __ballot(lane < nlmax && okrec && okidx && (X.pg || B.tok <= X.chi));
        const uint32_t nl_hard = ~mh ? (uint32_t)__ffsll((long long)~mh) - 1 : 64u, nl_res = ~mr ? (uint32_t)__ffsll((long long)~mr) - 1 : 64u;      // leading lanes that may go
        const bool drained = turn == ticket;                                            // every claimed batch is published
        uint32_t nl = nl_res;
        bool smode = false;                             // a lane whose regions do not fit the index ring at once: alone, filling the index as it goes
        if (!nl && nl_hard && drained) {
            // nothing in flight and the next row's bytes are not resident: the mover is about to bring them - or the row is
            // longer than the ring, then its lane reads from memory
            if (X.chi < csize && stall < 16) { stall++; parser_unlock(misc); spin_pause(); continue; }
            nl = 1;
        }
        if (!nl && !nl_hard && drained && nlmax && wave_readlane(okrec ? 1u : 0u, 0) && wave_readlane(Rb, 0) - Ra > kIdxRing / 2) { smode = true; nl = 1; }
        if (!nl) {
#ifdef LZ4AMD_PROF_PARSER
            n_nowork++;
#endif
            parser_unlock(misc); spin_pause_long(); PSTAMP(t_sel); continue; }
        stall = 0;
        const uint32_t RbN = wave_readlane(Rb, nl - 1), ordN = wave_readlane(B.ord, nl - 1);
        const bool allres = nl_res != 0;
        wave_lds_fence();
        if (lane == 0) {
            misc[P_NEXT] = r0 + nl; misc[P_RZ] = RbN > Ra ? RbN : Ra; misc[P_TICKET] = ticket + 1; misc[P_ORD] = ordN;
            lds_store_release(&misc[M_PR0], r0 + nl);       // (the rows of claimed batches are in their waves' registers)
        }
        if (!smode) parser_unlock(misc);                    // (a row that fills the index as it goes keeps the lock: nobody may run ahead of it)
        PSTAMP(t_sel);
        if (prof) { const uint64_t t = clock_ticks(); t_wait += t - tq; tq = t; n_batch++; n_lanes += nl; }
        if (!smode) { for (uint32_t R = Ra + lane; R < RbN; R += 64) idx[R & kIdxMask] = 0; }      // (notes below; regions without one are holes)
        wave_lds_fence();
        // ---- walk
        Walk w; w.p = A.tok; w.o = A.out + kBias; w.i = A.ord; w.end = B.tok; w.oend = B.out + kBias; w.iend = B.ord;
        w.act = lane < nl && w.p < w.end; w.res = allres; w.bad = false;
        w.W = X.pg ? g_fetch8(X.src, w.p, X.csize) : cr_fetch8(X.cr, w.p, X.mis);
        uint32_t head = 0, sRa = Ra, scarry = 0;
        while (__any(w.act)) {
            n_steps++;
            const uint32_t o0 = w.o, i0 = w.i;
            const StepOut S = parser_step(X, w, n_careful);
            if (S.ok) {
                recs[w.i & kRecMask] = S.r;
                const uint32_t gq = (w.o + kRegion - 1) >> kRegionShift;
                if (!smode && (gq << kRegionShift) < S.oe) idx[gq & kIdxMask] = w.i + 1;           // the first region whose first byte the sequence holds
            }
            w.i += S.ok ? 1u : 0u; w.o = S.oe;
            w.act = w.act && S.ok && w.p < w.end;
            if (smode) {
                // the index entries of this one sequence, as far as the ring has room, publishing as it goes (the copy must be
                // able to move on for room to appear).  Every earlier batch is published: the row is true, and so are its records.
                const uint32_t so = wave_readlane(o0, 0), se = wave_readlane(S.oe, 0), si = wave_readlane(i0, 0);
                if (wave_readlane(S.ok ? 1u : 0u, 0)) {
                    wave_lds_fence();
                    if (lane == 0) recs[(si + 1) & kRecMask].outpos = se;
                    head = si + 2;
                    uint32_t ga = (so + kRegion - 1) >> kRegionShift;
                    const uint32_t gb = (se + kRegion - 1) >> kRegionShift;
                    for (;;) {
                        wave_lds_fence();
                        if (lane == 0) lds_store_release64((uint64_t*)&misc[M_HEAD], (uint64_t)head | ((uint64_t)ga << 32));
                        if (ga >= gb) break;
                        const Ctl c2 = ctl_snapshot(smem);
                        if (c2.abort_) return;
                        const uint32_t n = gb - ga < 64 ? gb - ga : 64;
                        if (ga + n + 1 <= c2.open + kIdxRing) { if (lane < n) idx[(ga + lane) & kIdxMask] = si; ga += n; }
                        else spin_pause();
                    }
                    scarry = si + 1;
                    if (gb > sRa) sRa = gb;
                }
            }
        }
        if (prof) { const uint64_t t = clock_ticks(); t_walk += t - tq; tq = t; }
        // ---- every lane must have arrived exactly at the next row
        const bool arrived = lane >= nl || (!w.bad && w.p == w.end && w.o == w.oend && w.i == w.iend);
        const bool all_arrived = !__any(!arrived);
        const uint32_t iendN = wave_readlane(w.iend, nl - 1), oendN = wave_readlane(w.oend, nl - 1), tokN = wave_readlane(w.end, nl - 1);
        // ---- publish, once every earlier batch is published
        if (!smode) {
            for (;;) {
                uint32_t tn, ab;
                lds_load_2(&misc[P_TURN], &misc[M_ABORT], tn, ab);
                if (__builtin_amdgcn_readfirstlane(tn) == ticket) break;
                if (__builtin_amdgcn_readfirstlane(ab)) return;
                spin_pause();
            }
        }
        if (!all_arrived) { fail = true; if (smode) parser_unlock(misc); break; }
        wave_lds_fence();
#ifdef LZ4AMD_PROF_PARSER
        { const uint64_t t_ = clock_ticks(); t_turn += t_ - tp; tp = t_; t_turn -= 0; }
#endif
        uint32_t pubRa = RbN > Ra ? RbN : Ra;
        if (!smode) {
            // holes (regions whose first byte lies in a sequence that noted an earlier region) take the entry before them
            uint32_t icarry = uload(&misc[P_ICARRY]);
            for (uint32_t R = Ra; R < RbN; R += 64) {
                const uint32_t v = R + lane < RbN ? idx[(R + lane) & kIdxMask] : 0u;
                const uint32_t filled = wave_incl_max_u32(v > icarry ? v : icarry);
                icarry = wave_readlane(filled, 63);
                if (R + lane < RbN) idx[(R + lane) & kIdxMask] = filled - 1;
            }
            if (lane == 0) misc[P_ICARRY] = icarry;
        } else { if (lane == 0) misc[P_ICARRY] = scarry; if (sRa > pubRa) pubRa = sRa; }
        if (lane == 0) recs[iendN & kRecMask].outpos = oendN;           // where the batch's last record ends (only this field: the next batch may have written its first record there already - with this value)
        const bool final = r0 + nl == nreg;
        if (final) { if (lane == 0) idx[rend & kIdxMask] = nseq - 1; pubRa = rend + 1; }          // (the last region asks for the entry behind it like every other)
        wave_lds_fence();
        if (lane == 0) {
            lds_store_release64((uint64_t*)&misc[M_HEAD], (uint64_t)(iendN + 1) | ((uint64_t)pubRa << 32));
            lds_store_release(&misc[M_CLO], tokN);                       // (AFTER the records: the mover reads this word first, the heads second - a new
            lds_store_release(&misc[P_TURN], ticket + 1);                //  position never meets old heads, which would let go of literals still to be copied)
        }
        if (smode) { if (lane == 0) misc[P_RZ] = pubRa; parser_unlock(misc); }
    }
#ifdef LZ4AMD_PROF_PARSER
    // developer build: where the first parser wave's time goes (the walk's time is counted with the wait for its turn: subtract t_walk)
    if (prof && lane == 0 && wave_id() == kFirstParseWave) { prof[5] = (t_lock >> 4) | ((t_sel >> 4) << 32); prof[6] = (t_turn >> 4) | ((t_pub >> 4) << 32); prof[7] = n_nowork; }
#endif
#undef PSTAMP
    if (prof && lane == 0 && wave_id() == kFirstParseWave) { prof[2] = n_batch | ((uint64_t)n_lanes << 32); prof[3] = n_steps | ((uint64_t)n_careful << 32); prof[4] = (t_wait >> 4) | ((t_walk >> 4) << 32); }
    if (fail) {
        wave_lds_fence();
        if (lane == 0) { lds_store_release(&misc[M_PBAD], 1u); lds_store_release(&misc[M_ABORT], 1u); }
    }
}

// ------------------------------------------------------------------------------ COPY
struct RegionCtx {
    char* smem;
    uint32_t R, x0, x1, slot;       // region, its output range, its ring slot
    uint32_t g;                     // regions below g are final
    uint32_t chi;                   // compressed bytes resident
    uint32_t mis;                   // the compressed block's misalignment in memory (stream position P sits at ring address P + mis)
    uint32_t ringB;                 // output position of ring address 0 two laps below the region
    uint32_t j0, nrec;              // records that overlap the region
    uint32_t tagLo, tagHi;          // lap tags of the ring's two laps the region can read: the one below its own, its own
};

// Finality of output bytes is kept per 32-byte chunk of the output ring as ONE BYTE: the lap tag of the region whose
// bytes are final there (a region's lap = region / kSlots; tag = lap + 1 mod 256, 0 = nothing yet).  A slot is recycled
// only once nobody can read its old region (kMaxLead), so "the byte holds the tag of the lap I mean" is the whole test -
// no region numbers, no 64-bit masks, no first-open-region compare.
// (tags are 16 .. 255: a chunk with k pending pieces holds tag - k, see copy_region, and that must neither wrap nor be 0)
__device__ __forceinline__ uint32_t lap_tag(uint32_t lap) { return 16u + lap % 240u; }
// output bytes [sa, sb] final?  (sb - sa < 32; both within the 64 KB below the region's end)  oka / okb: the chunk of sa / of sb is
__device__ __forceinline__ void range_flags(const RegionCtx& C, uint32_t sa, uint32_t sb, bool& oka, bool& okb) {
    const uint8_t* done = (const uint8_t*)(C.smem + kOffBits);
    const uint32_t oa = sa - C.ringB, ob = sb - C.ringB;                 // [0, 2 * ring): the lap below the region's, then its own
    const bool ha = oa >= kRingBytes, hb = ob >= kRingBytes;
    uint32_t fa, fb;
    lds_load_flags2(done + ((ha ? oa - kRingBytes : oa) / kChunk), done + ((hb ? ob - kRingBytes : ob) / kChunk), fa, fb);
    oka = fa == (ha ? C.tagHi : C.tagLo); okb = fb == (hb ? C.tagHi : C.tagLo);
}
__device__ __forceinline__ bool range_is_final(const RegionCtx& C, uint32_t sa, uint32_t sb) {
    bool a, b; range_flags(C, sa, sb, a, b); return a && b;
}
__device__ __forceinline__ U32x8 ring_read32(const RegionCtx& C, uint32_t pos) {
    return lds_read_chunk_at((const uint8_t*)(C.smem + kOffRing), ring_fold(pos - C.ringB));
}

// What a piece's fetch found out.  ready: its bytes can be put now.  plain: they are v's bytes [lo, lo + n) (v holds the 32 bytes from the
// chunk's first byte on as the source has them); else the piece is a match whose period is shorter than the piece: it is copied
// byte by byte inside the ring, from dist bytes back (copy_short_period).
// key: a piece that is not ready for the most common reason - a plain match whose source bytes [key, key + n) lie in other chunks that
// are still in flight - reports where its source starts: it can be finished later with one poll of those chunks' lap tags and one ring
// read.  kKeyAlways: only a full attempt can tell.
enum : uint32_t { kKeyAlways = 0xFFFFFFFFu };
struct Fetch { U32x8 v; uint32_t key, dist; bool ready, plain; };
// Bytes [lo, lo + n) of a chunk := output bytes [d, d + n) of the piece (literal run or match of rec; ms = where the match starts).
// own_ok: every lower piece of d's own chunk is done (only a match with a period < 32 that starts inside a chunk reads its own chunk).
__device__ __forceinline__ Fetch item_fetch(const RegionCtx& C, bool is_lit, uint32_t d, uint32_t lo, uint32_t n,
                                            const SeqRec& rec, uint32_t ms, bool own_ok) {
    // Literal pieces and match pieces sit side by side in a wave, so both address computations run for every lane anyway:
    // they are written straight-line and end in ONE 32-byte LDS read at the selected address (compressed ring or output
    // ring) instead of one predicated read per kind.
    Fetch F; F.key = kKeyAlways;
    // -- a literal piece: stream bytes [A, A + n)
    const uint32_t A = rec.litpos + (d - rec.outpos);
    const uint32_t a = mod_cr(A + C.mis - lo);                               // (the ring is direct mapped on memory address: position + the block's misalignment; below position 0 only masked-off bytes)
    const bool lit_ok = A + n <= C.chi;
    // -- a match piece: output bytes [s, s + n)
    uint32_t dist = rec.off;
    const uint32_t into = d - ms;
#ifdef LZ4AMD_TRACE
    if (!is_lit && dist == 0) { fprintf(stderr, "ZERO OFFSET item: R=%u d=%u lo=%u n=%u rec{%u,%u,%u,%u} ms=%u j0=%u nrec=%u\n", C.R, d, lo, n, rec.outpos, rec.litpos, rec.ll, rec.off, ms, C.j0, C.nrec); }
#endif
    if (!is_lit && into >= dist) {
        // the source lies inside this very match (it overlaps itself): every earlier period holds the same bytes; read one
        // far back - the period times the largest power of two that stays inside the match's own source and the 64 KB
        // window: at least half as far as the farthest, long final - instead of the bytes just written (no division: two
        // of them were 70 instructions of this function)
        const uint32_t lim = umin32(into + dist, kMaxDistance);                  // dist << j <= lim
        uint32_t j = (uint32_t)__clz((int)dist) - (uint32_t)__clz((int)lim);     // floor(log2 lim) - floor(log2 dist) >= 0
        j -= ((dist << j) > lim) ? 1u : 0u;
        dist <<= j;
    }
    const uint32_t s = d - dist;
    // sources below my chunk must be final; sources inside my own chunk (a match that starts inside a chunk,
    // offset < 32 + lo) are in once every lower piece of the chunk is
    const uint32_t cstart = d - lo;
    const uint32_t se = n <= dist ? s + n - 1 : d - 1;                       // last source byte
    const bool below = s < cstart;
    bool src_final = true;
    if (!is_lit && below) src_final = range_is_final(C, s, se < cstart ? se : cstart - 1);
    const bool wait_own = src_final && se >= cstart && !own_ok;
    if (!is_lit && !src_final && n <= dist && se < cstart) F.key = s;
    F.ready = is_lit ? lit_ok : (src_final && !wait_own);
    F.plain = is_lit || n <= dist;
    F.dist = dist;
    {   // (read whether or not the piece is ready: the address is always inside its ring, and the read then does not wait
        //  for the lap tags' own trip to LDS)
        const uint32_t addr = is_lit ? kOffCr + a : kOffRing + ring_fold(s - lo - C.ringB);
        F.v = lds_read_chunk_at((const uint8_t*)C.smem, addr);
    }
    return F;
}

__device__ __forceinline__ void lds_or32(char* smem, uint32_t off, const U32x8& v) {
    unsigned long long* q = (unsigned long long*)(smem + off);
    atomicOr(&q[0], (unsigned long long)v.a[0] | ((unsigned long long)v.a[1] << 32));
    atomicOr(&q[1], (unsigned long long)v.a[2] | ((unsigned long long)v.a[3] << 32));
    if constexpr (kWide) {
        atomicOr(&q[2], (unsigned long long)v.b[0] | ((unsigned long long)v.b[1] << 32));
        atomicOr(&q[3], (unsigned long long)v.b[2] | ((unsigned long long)v.b[3] << 32));
    }
}

// chunk c of the region's slot |= v (the ring's pad mirrors chunks 0-1 of slot 0 at all times, so that a chunk-sized
// read that starts in the ring's last bytes runs on into valid data)
__device__ __forceinline__ void slot_or32(const RegionCtx& C, uint32_t c, const U32x8& v) {
    lds_or32(C.smem, kOffRing + (C.slot << kRegionShift) + c * kChunk, v);
    if (C.slot == 0 && c < kRingPad / kChunk) lds_or32(C.smem, opaque_u32(kOffRing + kRingBytes) + c * kChunk, v);      // (opaque: or the compiler keeps the four addresses in registers through every loop around)
}
__device__ __forceinline__ void slot_write32(const RegionCtx& C, uint32_t c, const U32x8& v) {
    U32x4* q = (U32x4*)(C.smem + kOffRing + (C.slot << kRegionShift) + c * kChunk);
    q[0] = v.a; if constexpr (kWide) q[1] = v.b;
    if (C.slot == 0 && c < kRingPad / kChunk) { U32x4* m = (U32x4*)(C.smem + opaque_u32(kOffRing + kRingBytes) + c * kChunk); m[0] = v.a; if constexpr (kWide) m[1] = v.b; }
}
// A match whose period is shorter than the piece (dist < n <= 32; only the first chunk or two of a match with a short period: deeper
// into it an earlier, farther period is read): output bytes [d, d + n) := the bytes dist before them, one after the other, inside the
// ring - the serial copy of the format, by one lane (the DS unit serves a wave's LDS operations in order).  The bytes are the piece's
// own: nobody else writes them, what the other pieces of the chunk OR in there is zero.
__device__ __forceinline__ void copy_short_period(const RegionCtx& C, uint32_t d, uint32_t dist, uint32_t n) {
    uint8_t* ring = (uint8_t*)(C.smem + kOffRing);
    const uint32_t od = (C.slot << kRegionShift) + (d - C.x0);             // ring address of d (inside the region's slot)
#pragma nounroll
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t b = lds_load_byte(ring + ring_fold(d + i - dist - C.ringB));
        lds_store_byte(ring + od + i, b);
        if (od + i < kRingPad) lds_store_byte(ring + kRingBytes + od + i, b);      // (slot 0's first bytes: the mirror behind the ring's end)
    }
}

// chunks that still have a pending piece: OR of 1 << chunk over the lanes of pm (few lanes: a scalar loop)
__device__ __forceinline__ uint64_t chunks_of(unsigned long long pm, uint32_t chunk) {
    uint64_t m = 0;
    while (pm) {
        const uint32_t l = (uint32_t)__ffsll((long long)pm) - 1; pm &= pm - 1;
        m |= 1ull << wave_readlane(chunk, l);
    }
    return m;
}

// round-B item of lane l in trip t: the head of a piece that starts inside a chunk
struct BItem { SeqRec rec; uint32_t ms, d, lo, n, chunk; bool is_lit, valid; };
__device__ __forceinline__ BItem b_item(const RegionCtx& C, uint32_t t) {
    const SeqRec* recs = (const SeqRec*)(C.smem + kOffRecs);
    const uint32_t l = lane_here(), r = 32 * t + (l >> 1);
    BItem it; it.valid = false; it.is_lit = !(l & 1);
    it.rec.outpos = it.rec.litpos = it.rec.ll = it.rec.off = 0; it.ms = it.d = it.lo = it.n = it.chunk = 0;
    if (r < C.nrec) {
        it.rec = recs[(C.j0 + r) & kRecMask];
        const uint32_t nout = recs[(C.j0 + r + 1) & kRecMask].outpos;
        it.ms = it.rec.outpos + it.rec.ll;
        it.d = it.is_lit ? it.rec.outpos : it.ms;
        const uint32_t pe = it.is_lit ? it.ms : nout;
        it.lo = it.d & (kChunk - 1);
        it.valid = it.d >= C.x0 && it.d < C.x1 && it.lo != 0 && pe > it.d;
        const uint32_t n = pe - it.d;
        it.n = n < kChunk - it.lo ? n : kChunk - it.lo;
        it.chunk = (it.d - C.x0) / kChunk;
    }
    return it;
}
// what the retry loop keeps of a pending round-B item of the first trip: lo (5 bits), n (6), chunk (6)
__device__ __forceinline__ uint32_t pack_b(uint32_t lo, uint32_t n, uint32_t chunk) { return lo | (n << 5) | (chunk << 11); }
__device__ __forceinline__ uint32_t pack_lo(uint32_t p) { return p & 31u; }
__device__ __forceinline__ uint32_t pack_n(uint32_t p) { return (p >> 5) & 63u; }
__device__ __forceinline__ uint32_t pack_chunk(uint32_t p) { return p >> 11; }

// Pending pieces per chunk: a byte per chunk, four to a word (a 32-byte chunk starts at most 13 pieces).
__device__ __forceinline__ uint32_t cnt_one(uint32_t c) { return 1u << (8u * (c & 3u)); }
__device__ __forceinline__ uint32_t cnt_of(uint32_t word, uint32_t c) { return (word >> (8u * (c & 3u))) & 0xFFu; }
// A chunk's flag byte holds the region's lap tag when the chunk is final, and tag - k while k of its pieces are pending (anything but the
// tag is "not final" to a reader).  A pending piece that is in - its bytes were written BEFORE: the DS unit serves a wave's operations in
// order - adds one to the byte: the chunk's last piece makes it final, no lane has to know that it was the last, nothing is read back.
__device__ __forceinline__ void piece_landed(uint8_t* done, uint32_t c) { c = opaque_u32(c); atomicAdd((uint32_t*)done + (c >> 2), cnt_one(c)); }      // (opaque: the address is made here, not kept in a register - or in scratch - through the loops around)

// What a waiting piece looks at: the flag bytes of its source's first and last chunk (LDS offsets) and the tags they must hold.
struct Watch { uint32_t f0, f1, tags; };
__device__ __forceinline__ Watch watch_of(const RegionCtx& C, uint32_t sa, uint32_t sb) {
    const uint32_t oa = sa - C.ringB, ob = sb - C.ringB;                 // [0, 2 * ring): the lap below the region's, then its own
    const bool ha = oa >= kRingBytes, hb = ob >= kRingBytes;
    Watch W;
    W.f0 = kOffBits + (ha ? oa - kRingBytes : oa) / kChunk; W.f1 = kOffBits + (hb ? ob - kRingBytes : ob) / kChunk;
    W.tags = (ha ? C.tagHi : C.tagLo) | ((hb ? C.tagHi : C.tagLo) << 8);
    return W;
}

// Compose region C.R in its ring slot and store it.
//   first pass   round A (lane = chunk: the piece that covers the chunk's first byte) and round B (lane = piece head: every piece that starts
//                inside a chunk, 32 records a trip).  A piece whose source is not final yet (it lies in a region another wave is composing) is
//                PENDING: counted on its chunk.  Right behind the pass every chunk's flag byte is set: the tag, less its pending pieces.
//   landing      every waiting lane looks at the flag bytes of its own source chunks; a piece that is in is ORed into its chunk and counted
//                on the chunk's flag byte, and the slot's BELL (kOffProg) moves.  Nothing came in: the wave sleeps on the bell of the region
//                that holds the highest chunk it waits for.  From one region's piece to the piece of the next region that waited for it
//                that is: add to the flag byte -> bell -> wake: bell and flags in one trip -> ring read -> OR, add -> bell.
//                (round 5: the waiting wave slept until the WHOLE source region was complete and went round a loop of ~10 dependent trips
//                to the LDS and ~300 instructions before its own chunks were flagged: with most regions waiting for a piece of a region just
//                below, that chain set the decoder's pace.)
enum : uint32_t { kKeyed = 3 };
static_assert(kKeyed == 3, "lds_load_word_then6");                 // waits a lane keeps in registers: its round-A piece, its pieces of round B's first two trips
__device__ __forceinline__ void copy_region(RegionCtx& C, lz4amd_gdst dst, uint32_t w, uint64_t* prof, uint64_t& t_retry, uint32_t& n_iter) {
    char* smem = C.smem;
    const bool timed = prof != nullptr;
    DSTAMP_DECL; DSTAMP(0);                             // first pass begins
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    const SeqRec* recs = (const SeqRec*)(smem + kOffRecs);
    uint8_t* done = (uint8_t*)(smem + kOffBits) + C.slot * 64;        // my chunks' bytes (they hold the tag of the lap below: not final)
    unsigned long long* pend = (unsigned long long*)(smem + kOffPend + w * kPendStride);     // pending lanes of round B, per trip
    uint32_t* cnt = (uint32_t*)(pend + kMaxTrips);                    // pending pieces per chunk
    uint32_t* bells = (uint32_t*)(smem + kOffProg);
    uint32_t mybell = C.R << 8;                                       // the slot's bell: every lane stores the same next value (values no earlier region left in this slot)
    const uint32_t lane = lane_here();
    const uint32_t slot_off = kOffRing + (C.slot << kRegionShift);
    {   const uint32_t lap = (C.R - C.slot) / kSlots; C.tagHi = lap_tag(lap); C.tagLo = lap_tag(lap - 1); }
    C.ringB = (C.R - C.slot - kSlots) << kRegionShift;
    // ---- round A: which record covers the first byte of each chunk?  (scratch: the slot itself)
    uint32_t* fs = (uint32_t*)(smem + slot_off);
    fs[lane] = 0;                                      // (cnt[] is all zero here: a region that counted pending pieces clears it behind its first pass)
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
        an = pe - c0 < kChunk ? pe - c0 : kChunk;
    }
    // what my pending pieces wait for: [0] round A, [1] round B's first trip.  key: where the source starts (kKeyAlways: not a
    // plain wait for source chunks - only a full attempt can tell); pack: lo / n / chunk
    uint32_t key[kKeyed] = {kKeyAlways, kKeyAlways}, pack[kKeyed] = {0, 0};
    uint64_t pm[kKeyed] = {0, 0};                    // pending lanes of the three
    uint32_t npend;                                     // pending pieces of the region
    bool later = false;                                 // ... some of them in round B's trips behind the first
    {
        const BItem it0 = b_item(C, 0);                // round B's first trip reads its records while round A's reads are in flight
        bool readyA = false;
        if (actA) {
            const U32x8 mA = mask_row(smem, an);                   // (read before the piece's source: one trip to the LDS for both)
            const Fetch F = item_fetch(C, alit, c0, 0, an, arec, ams, true);
            readyA = F.ready; key[0] = F.key; pack[0] = pack_b(0, an, lane);
            slot_write32(C, lane, F.ready && F.plain ? and_row(F.v, mA) : zero8());
            if (F.ready && !F.plain) copy_short_period(C, c0, F.dist, an);
            if (F.ready && an == kChunk) lds_store_flag(done + lane, C.tagHi);      // (the chunk is this one piece: final here and now)
            if (!F.ready) atomicAdd(&cnt[lane >> 2], cnt_one(lane));
        }
        pm[0] = __ballot(actA && !readyA);
        npend = (uint32_t)__popcll(pm[0]);
        // ---- round B: heads of the pieces that start inside a chunk, 32 records per trip
        const uint32_t trips = (C.nrec + 31) / 32;
        for (uint32_t t = 0; t < trips; t++) {
            const BItem it = t == 0 ? it0 : b_item(C, t);
            bool rdy = false;
            if (it.valid) {
                const U32x8 mh = mask_row(smem, it.lo + it.n), ml = mask_row(smem, it.lo);
                const Fetch F = item_fetch(C, it.is_lit, it.d, it.lo, it.n, it.rec, it.ms, false);
                rdy = F.ready;
                if (rdy) { if (F.plain) slot_or32(C, it.chunk, and_rows(F.v, mh, ml)); else copy_short_period(C, it.d, F.dist, it.n); }
                else {
                    atomicAdd(&cnt[it.chunk >> 2], cnt_one(it.chunk));
                    if (t == 0) { key[1] = F.key; pack[1] = pack_b(it.lo, it.n, it.chunk); }
                }
            }
            const unsigned long long m = __ballot(it.valid && !rdy);
            if (lane == 0) pend[t] = m;
            npend += (uint32_t)__popcll(m);
            if (t == 0) pm[1] = m; else later = later || m != 0;
        }
        // ---- every chunk's flag byte: the tag less the chunk's pending pieces (none: final) - tell the other waves
        wave_lds_fence();                              // (the pieces' bytes and the counts first)
        if (npend == 0) lds_store_flag(done + lane, C.tagHi);
        else {
            lds_store_flag(done + lane, C.tagHi - cnt_of(cnt[lane >> 2], lane));
            wave_lds_fence();
            if (lane < 16) cnt[lane] = 0;              // (for the wave's next region)
        }
    }
    DSTAMP(1);                                          // first pass done
    const uint32_t npend0 = npend; (void)npend0;
    // ---- pieces whose sources were still in flight: land them as their sources come in
    bool aborted = false;                               // the block was given up (M_ABORT) while this region waited
    if (npend) {
        const uint64_t tr0 = timed ? clock_ticks() : 0;
        const uint32_t trips = (C.nrec + 31) / 32;
        uint32_t idle_polls = 0;
        wave_lds_order();
        { lds_store_relaxed(&bells[C.slot], ++mybell); wake_workgroup(); }
        const uint32_t* const still = &misc[M_SPARE];  // (a word that does not move: the bell of a lane that watches none)
        const uint32_t* bp[kKeyed] = {still, still};   // the bells my two waits watch: those of the regions that hold the chunks they wait for
        bool dirty = true;                             // the waits changed: look again at what the lanes watch
        Watch W[kKeyed];
        bool on[kKeyed] = {false, false};
        bool hard = false;
        for (;;) {                                     // (left by break only: a return out of the nest of loops is a construct that has miscompiled on the device before)
            n_iter++;
            if (dirty) {
                hard = later;
#pragma unroll
                for (uint32_t k = 0; k < kKeyed; k++) {
                    const bool mine = (pm[k] >> lane) & 1ull;
                    on[k] = mine && key[k] != kKeyAlways;
                    hard = hard || __any(mine && key[k] == kKeyAlways);
                    W[k] = watch_of(C, key[k], key[k] + pack_n(pack[k]) - 1);
                    if (!on[k]) { W[k].f0 = W[k].f1 = kOffBits; bp[k] = still; }
                }
                dirty = false;
            }
            // -- one trip to the LDS: my bells, THEN the flag bytes my pieces wait for (a flag that is set behind this look moves its
            //    region's bell behind the value read here)
            uint32_t seen[kKeyed], fl[2 * kKeyed];
            {
                const uint8_t* sm8 = (const uint8_t*)smem;
                const uint8_t* const fp[4] = {sm8 + W[0].f0, sm8 + W[0].f1, sm8 + W[1].f0, sm8 + W[1].f1};
                lds_load_words_then4(bp[0], bp[1], fp, seen[0], seen[1], fl);
                wave_converge();
            }
            bool in[kKeyed], ok0[kKeyed];
            bool any_in = false;
#pragma unroll
            for (uint32_t k = 0; k < kKeyed; k++) {
                ok0[k] = fl[2 * k] == (W[k].tags & 0xFFu);
                in[k] = on[k] && ok0[k] && fl[2 * k + 1] == (W[k].tags >> 8);
                any_in = any_in || in[k];
            }
            bool progress = false;
            if (__any(any_in)) {
                uint32_t landed = 0;
#pragma unroll
                for (uint32_t k = 0; k < kKeyed; k++) {
                    if (in[k]) {
                        const uint32_t lo = pack_lo(pack[k]), n = pack_n(pack[k]), ch = pack_chunk(pack[k]);
                        slot_or32(C, ch, keep_bytes(smem, ring_read32(C, key[k] - lo), lo, lo + n));
                        piece_landed(done, ch);
                    }
                    const unsigned long long dn = __ballot(in[k]);
                    if (dn) { pm[k] &= ~dn; landed += (uint32_t)__popcll(dn); if (k && lane == 0) pend[k - 1] = pm[k]; }
                }
                npend -= landed;
                wave_lds_order();
                { lds_store_relaxed(&bells[C.slot], ++mybell); wake_workgroup(); }
                DSTAMP(3); DSTAMP_LAST(4);     // first / last landing the cheap way
                if (!npend) break;
                progress = true; dirty = true;
            }
            // -- the other waits (literals whose stream bytes are not resident yet, matches that read their own chunk, matches with a period
            //    shorter than the piece, everything of the later trips): the full attempt, when the cheap way brought nothing
            if (hard && !progress) {
                {   const Ctl c = ctl_snapshot(smem);          // literal pieces wait for stream bytes - how many are resident now?
                    if (c.abort_) { aborted = true; break; }
                    C.chi = c.chi; C.g = c.open; }
                wave_lds_fence();                              // (pend[] of the last pass)
                uint32_t landed = 0;
                if (pm[0]) {
                    const bool mine = (pm[0] >> lane) & 1ull;
                    bool rdy = false;
                    if (mine) {
                        const Fetch F = item_fetch(C, alit, c0, 0, an, arec, ams, true);
                        rdy = F.ready; key[0] = F.key;
                        if (rdy) { if (F.plain) slot_or32(C, lane, keep_low_bytes(smem, F.v, an)); else copy_short_period(C, c0, F.dist, an);
                                   piece_landed(done, lane); }
                    }
                    const unsigned long long dn = __ballot(rdy);
                    pm[0] &= ~dn; landed += (uint32_t)__popcll(dn);
                }
                bool earlier_clear = true;
                later = false;
                for (uint32_t t = 0; t < trips; t++) {
                    const unsigned long long m = pend[t];
                    if (!m) continue;
                    const BItem it = b_item(C, t);
                    const bool mine = (m >> lane) & 1ull;
                    // is every lower piece of my own chunk in? (chunk A, the earlier trips, the lower lanes of this trip)
                    const unsigned long long lower = m & ((1ull << lane) - 1ull);
                    const uint32_t h = lower ? 63u - (uint32_t)__clzll((long long)lower) : 0u;
                    const uint32_t hc = (uint32_t)__shfl((int)it.chunk, (int)h);
                    const bool own_ok = earlier_clear && !((pm[0] >> it.chunk) & 1ull) && (!lower || hc != it.chunk);
                    bool rdy = false;
                    if (mine) {
                        const Fetch F = item_fetch(C, it.is_lit, it.d, it.lo, it.n, it.rec, it.ms, own_ok);
                        rdy = F.ready;
                        if (!rdy && t == 0) { key[1] = F.key; pack[1] = pack_b(it.lo, it.n, it.chunk); }
                        if (rdy) { if (F.plain) slot_or32(C, it.chunk, keep_bytes(smem, F.v, it.lo, it.lo + it.n)); else copy_short_period(C, it.d, F.dist, it.n);
                                   piece_landed(done, it.chunk); }
                    }
                    const unsigned long long dn = __ballot(rdy), left = m & ~dn;
                    landed += (uint32_t)__popcll(dn);
                    if (lane == 0) pend[t] = left;
                    if (t == 0) pm[1] = left; else later = later || left != 0;
                    if (left) earlier_clear = false;
                }
                npend -= landed;
                DSTAMP(5);                              // first full attempt
                dirty = true;
                if (landed) { wave_lds_order(); { lds_store_relaxed(&bells[C.slot], ++mybell); wake_workgroup(); } }
                if (!npend) break;
                progress = landed != 0;
            }
            if (progress) { idle_polls = 0; continue; }
            // -- nothing came in.  Every waiting lane watches the bell of the region that holds the chunk it waits for (its source's first
            //    chunk that is not final; a chunk of my own region: none, it follows from my other pieces) - if that is the bell it read
            //    before the flags; else the wave looks once more, bells first.  Then it sleeps until somebody's s_wakeup, and goes on when
            //    one of its lanes' bells has moved.
            bool moved = false;
#pragma unroll
            for (uint32_t k = 0; k < kKeyed; k++) {
                const uint32_t* nb = still;
                if (on[k]) { const uint32_t r = (ok0[k] ? key[k] + pack_n(pack[k]) - 1 : key[k]) >> kRegionShift; if (r < C.R) nb = &bells[r % kSlots]; }
                moved = moved || nb != bp[k];
                bp[k] = nb;
            }
            if (__any(moved)) continue;
            if (__any(bp[0] != still || bp[1] != still)) {
                for (uint32_t k = 0; k < 24; k++) {            // (bounded: a lost wake-up, a wait that is not what it seemed)
                    uint32_t b0, b1;
                    lds_load_2v(bp[0], bp[1], b0, b1);
                    if (__any(b0 != seen[0] || b1 != seen[1])) break;
                    if ((k & 3u) == 3u && uload(&misc[M_ABORT])) { aborted = true; break; }
                    sleep_until_woken(); DCOUNT();
                }
                DSTAMP(2);                                  // first time the sleep ended
            } else { if (++idle_polls > 2) spin_pause_long(); else spin_pause(); }
            if (aborted || uload(&misc[M_ABORT])) { aborted = true; break; }
        }
        if (timed) t_retry += clock_ticks() - tr0;
    }
    if (!aborted) {
        // ---- the region is complete (every chunk is flagged): its bytes into registers, tell the other waves (they read the ring, not HBM: a
        //      region that waits for this one need not wait for the store as well), then to HBM - in rows of 64 aligned 16-byte
        //      stores (lane l: bytes [16 l, 16 l + 16) of either KB of the region): 1 KB contiguous per store instruction
        wave_lds_fence();
        const uint32_t h0 = C.x0 + 16 * lane, h1 = h0 + 1024;
        U32x4 v0, v1; v0[0] = v0[1] = v0[2] = v0[3] = 0; v1 = v0;
        if (h0 < C.x1) v0 = *(const U32x4*)(smem + slot_off + 16 * lane);
        if constexpr (kWide) { if (h1 < C.x1) v1 = *(const U32x4*)(smem + slot_off + 1024 + 16 * lane); }
        wave_lds_fence();                                   // (the slot is not read again: whoever recycles it may)
        if (lane == 0) lds_store_release((uint32_t*)(smem + kOffRegDone) + C.slot, C.R + 1);
        lds_store_relaxed(&bells[C.slot], ++mybell);
        wake_workgroup();
        DSTAMP(6); DFLUSH(prof, C.R, npend0 > 255 ? 255u : npend0);      // complete
        if (h0 < C.x1) {
            if (h0 + 16 <= C.x1) st_global16(dst + (h0 - kBias), v0);
            else {
#pragma nounroll
                for (uint32_t i = 0; i < C.x1 - h0; i++) dst[h0 - kBias + i] = (uint8_t)chunk_byte(v0, i);
            }
        }
        if constexpr (kWide) {
            if (h1 < C.x1) {
                if (h1 + 16 <= C.x1) st_global16(dst + (h1 - kBias), v1);
                else {
#pragma nounroll
                    for (uint32_t i = 0; i < C.x1 - h1; i++) dst[h1 - kBias + i] = (uint8_t)chunk_byte(v1, i);
                }
            }
        }
    }
}

__device__ __forceinline__ void copy_role(uint32_t w, lz4amd_gsrc src, lz4amd_gdst dst, uint32_t nseq, uint32_t total, uint32_t rend, char* smem, uint64_t* prof, bool hinted) {
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    const uint32_t lane = lane_here();
    uint32_t k = 0;
    uint64_t t_rec = 0, t_lead = 0, t_work = 0, t_retry = 0;
    uint32_t n_iters = 0, n_retried = 0, n_lead = 0, n_cov = 0;
    for (;;) {
        // regions are handed out in order to whichever wave is free: a slow region does not hold up its wave's next ones
        // (asking for the next region before the current one's last steps, to hide the counter's latency, measured slower: 1.86 -> 1.89 ms)
        uint32_t R = 0;
        if (lane == 0) R = atomicAdd(&misc[M_NEXT], 1u);
        R = __builtin_amdgcn_readfirstlane(R);
        if (R >= rend) break;
        const uint32_t slot = R % kSlots;
        RegionCtx C; C.smem = smem; C.R = R; C.slot = slot; C.mis = stream_misalign(src);
        C.x0 = R << kRegionShift;
        C.x1 = C.x0 + kRegion < kBias + total ? C.x0 + kRegion : kBias + total;
        // ---- wait until the region's records are in the ring and its ring slot is free: region R takes the slot of
        //      region R-80, which regions up to R-16 may still read
        uint64_t ts = prof ? clock_ticks() : 0;
        const uint32_t iwant = (hinted || R + 2 < rend) ? R + 2 : rend;        // index entries of R and R + 1 (the parser writes one behind the last region)
        for (;;) {
            uint32_t i0, i1;
            const Ctl c = ctl_snapshot_region(smem, R, i0, i1);
            C.g = c.open; C.chi = c.chi;
            if (c.abort_) goto out;
            bool covered = c.ihead >= iwant;
            if (covered) {
                C.j0 = i0;
                const uint32_t jl = (hinted || R + 1 < rend) ? i1 : nseq - 1;
                uint32_t nrec = jl - C.j0 + 1;
                if (nrec > 32 * kMaxTrips) nrec = 32 * kMaxTrips;             // (more than 258 records never overlap a region)
                C.nrec = nrec;
                covered = c.head >= jl + 2;                                    // (row jl + 1 says where record jl ends)
            }
            if (covered && C.g + kMaxLead >= R) break;
            if (covered) { n_lead++; spin_pause(); } else { n_cov++; spin_pause_long(); }      // (a wave that waits for records leaves the issue slots to the waves that make them)
        }
        if (prof) {   // (the clock reads and this bookkeeping only under the developer profile: they were ~5 % of a region's time)
            const uint64_t t = clock_ticks(), dt = t - ts;
            if (n_lead > n_cov) t_lead += dt; else t_rec += dt;
            ts = t; n_lead = n_cov = 0; }
        uint64_t tr = 0; uint32_t ni = 0;
        DTRACE("region R=%u x0=%u x1=%u j0=%u nrec=%u g=%u\n", R, C.x0, C.x1, C.j0, C.nrec, C.g);
        copy_region(C, dst, w, prof, tr, ni);
        n_iters += ni; n_retried += ni ? 1u : 0u;
        DTRACE("region R=%u done\n", R);
        k++;
        // (the first-open-region word is moved by the mover wave: it reads the complete marks of the regions in flight in one trip)
        if (prof) { const uint64_t t = clock_ticks(); t_work += t - ts - tr; t_retry += tr; }
    }
out: ;
#ifndef LZ4AMD_PROF_PARSER
    if (prof && w == 0 && lane == 0) { prof[6] = t_rec | (t_lead << 32); prof[7] = t_work | (t_retry << 32); prof[5] = n_iters | ((uint64_t)n_retried << 32) | ((uint64_t)k << 48); }
#endif
}

// ------------------------------------------------------------------------------ stage B of one block
// hint: the block's entry-point table (null: records and region index come from stage A's scratch).  Returns false when
// the table turned out not to fit the stream (the block must then be decoded again without it).
__device__ __forceinline__ bool stream_block(lz4amd_gsrc src, uint32_t csize, lz4amd_gdst dst, uint32_t cap, uint32_t prefix,
                                             const SeqRec* rectab, const uint32_t* ridx, uint32_t nseq, uint32_t total, char* smem, uint64_t* prof,
                                             lz4amd_gsrc hint, uint32_t nreg) {
    const uint32_t tid = threadIdx.x, w = wave_id();
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    // ---- stage B: control words, done entries, the history before dst (linked blocks, lz4.c:2719 usingDict prefix mode) -> ring
    if (tid == 0) { misc[M_ABORT] = 0; misc[M_SPARE] = 0; misc[M_CHI] = 0; misc[M_IHEAD] = kFirstRegion; misc[M_HEAD] = 0; misc[M_CLO] = 0; misc[M_NEXT] = kFirstRegion; misc[M_OPEN] = kFirstRegion;
                    misc[M_EHEAD] = 0; misc[M_PR0] = 0; misc[M_PBAD] = 0;
                    misc[P_LOCK] = 0; misc[P_NEXT] = 0; misc[P_RZ] = kFirstRegion; misc[P_TICKET] = 0; misc[P_TURN] = 0; misc[P_ICARRY] = 0; misc[P_ORD] = 0; }
    if (tid < 2 * (kChunk + 1)) { U32x4 m; for (uint32_t k = 0; k < 4; k++) m[k] = low_bytes_mask(tid >> 1, 4 * (tid & 1) + k); *(U32x4*)(smem + kOffMaskTab + 16 * tid) = m; }
    // chunk flags: the history before dst (positions below kBias = regions 0..63, lap 0) is final, nothing else is
    for (uint32_t i = tid; i < kSlots * 16; i += kDecThreads) ((uint32_t*)(smem + kOffBits))[i] = i < kFirstRegion * 16 ? 0x01010101u * lap_tag(0) : 0u;
    if (tid < kSlots) { ((uint32_t*)(smem + kOffRegDone))[tid] = 0; ((uint32_t*)(smem + kOffProg))[tid] = 0; }
    for (uint32_t i = tid; i < kActiveCopy * kPendStride / 4; i += kDecThreads) ((uint32_t*)(smem + kOffPend))[i] = 0;      // (the copy waves' pending counts start at zero)
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

    const uint32_t rend = (kBias + total + kRegion - 1) >> kRegionShift;          // regions [kFirstRegion, rend)
    const bool hinted = hint != nullptr;
    // (nreg: rows of the table; nreg + 1 entries behind its 16-byte header)
    if (w == kMoveWave) mover_role(src, csize, rectab, ridx, nseq, rend, smem, hinted ? hint + LZ4AMD_HINT_HEAD : hint, nreg + 1);
    else if (hinted && w >= kFirstParseWave) parser_role(src, csize, cap, prefix, total, nseq, nreg, rend, smem, prof);
    else if (w < kActiveCopy) copy_role(w, src, dst, nseq, total, rend, smem, prof, hinted);
    __syncthreads();
    return misc[M_PBAD] == 0;
}

// ------------------------------------------------------------------------------ one DEPENDENT block
// lz4frame's linked blocks (lz4frame.c:1901-1915): block b's output starts where block b-1's ended and its matches may
// reach 64 KB back into it.  All blocks of the frame are in ONE launch: a workgroup pre-parses its block at once -
// against the largest history there can be - then waits for the workgroup that owns b-1 to publish where its output
// ended (tickets are handed out in block order to running workgroups, so that owner exists and progresses), checks the
// lowest position its matches refer to against the history that is really there, and streams.  Only stage B is serial.
__device__ __forceinline__ void chain_publish(const DecBatch& P, uint32_t b, long long v) {
    __syncthreads();                                       // every store of the block was issued
    if (threadIdx.x == 0) chain_store_release(&P.chain[b + 1], v);
}
__device__ __forceinline__ bool chain_run_head(const DecBatch& P, uint32_t b) { return b == 0 || (P.stored && (P.stored[b] & 2)); }
__device__ __forceinline__ bool chain_run_tail(const DecBatch& P, uint32_t b) { return b + 1 >= P.n_blocks || (P.stored && (P.stored[b + 1] & 2)); }
// ------------------------------------------------------------------------------ one block
// (independent and dependent blocks share ONE copy of stage A and stage B: the kernel is instruction-cache bound enough)
// use_hints: try the block's entry-point table, if the plan has one.  Returns false when the table did not fit the stream:
// the caller decodes the block again without it.
// kRuns: the kernel of dependent blocks (lz4amd_k_decompress_runs: gated blocks, lowref).  The general kernel (lz4amd_k_decompress: the step's) is,
// instruction for instruction, what it was before there were gated blocks.  Both ask P.chain at run time: a build in which kRuns made `chained` a
// compile-time constant (either way: no chain code in the one, no tables' code in the other; 204 / 147 spilled scalars instead of 232) faulted in
// its first launch, with or without -amdgpu-spill-sgpr-to-vgpr=false (not understood; this form is the one that is tested).
constexpr uint32_t kNoTwin = 0xFFFFFFFFu;
template <bool kRuns>
__device__ __forceinline__ bool decode_one_block(const DecBatch& P, uint32_t b, char* smem, bool use_hints, uint32_t twin_of = kNoTwin) {
    const uint32_t tid = threadIdx.x;
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);

    const lz4amd_gsrc src = LZ4AMD_TO_GSRC(P.src[b]);
    lz4amd_gdst dst = LZ4AMD_TO_GDST(P.dst[b]);
    const int32_t csize_i = P.src_size[b];
    const int32_t cap_i = P.dst_cap[b];
    const bool chained = P.chain != nullptr;
    const bool stored = chained && P.stored && (P.stored[b] & 1);
    // a launch may hold several chains ("runs": kernels/chain_spec_kernel.h decodes stretches of a linked frame side by side): bit 1 of a block's
    // flag marks the first block of a run - it starts at its own dst[b] with prefix[b] bytes of history and waits for nobody; the blocks
    // behind it have the same dst[] and prefix[] and count their positions from there
    // (both questions are asked where they are needed, not kept in registers across the block's decode: the kernel spills scalars as it is)

    // -- degenerate inputs (lz4.c:2036, 2062-2069); in a chain they are failures like any other
    bool ok = true;
    if (!chained) {
        if (src == nullptr || cap_i < 0) { if (tid == 0) P.result[b] = -1; return true; }
        if (cap_i == 0) {
            if (tid == 0) P.result[b] = (csize_i == 1 && src[0] == 0) ? 0 : -1;
            return true;
        }
        if (csize_i <= 0) { if (tid == 0) P.result[b] = -1; return true; }
    } else ok = src != nullptr && (csize_i > 0 || (stored && csize_i == 0)) && cap_i > 0 && !(stored && csize_i > cap_i);    // (a stored block may be empty: tests/frametest.c:1237 inserts such blocks)
    const uint32_t csize = (uint32_t)csize_i, cap = (uint32_t)cap_i;
    // a dependent block is pre-parsed against the largest history there can be; what is really there is checked below
    uint32_t prefix = chained ? kBias : (P.prefix ? (uint32_t)P.prefix[b] : 0u); if (prefix > kBias) prefix = kBias;

    SeqRec* rectab = (SeqRec*)(P.scratch + (uint64_t)blockIdx.x * P.scratch_stride);
    uint64_t* prof = (LZ4AMD_DEC_PROF && P.prof) ? P.prof + (uint64_t)blockIdx.x * 8 : nullptr;      // (a developer build's stamps: the product kernel carries none of that code)
    uint64_t tstart = 0;
    if (prof && tid == 0) tstart = clock_ticks();

    // ---- a block that comes with an entry-point table needs no stage A: the table's header says how much output and how
    //      many sequences to expect; whether the table tells the truth is found out while it is used (PARSER)
    uint32_t nseq = 0, total = stored ? csize : 0u;
    uint32_t* ridx = nullptr;
    lz4amd_gsrc hint = nullptr;
    uint32_t nrows = 0;
    if (use_hints && !chained && P.hints && P.hint_stride >= 48) {
        const lz4amd_gsrc hp = LZ4AMD_TO_GSRC(P.hints + (uint64_t)b * P.hint_stride);
        const U32x4 h = ld_global16(hp);                     // { magic, output bytes, compressed bytes, sequences }
        const U32x4 e0 = ld_global16(hp + 16);               // { rows, 0, 0, 0 }
        if (h[0] == LZ4AMD_HINT_MAGIC && h[2] == csize && csize < LZ4AMD_HINT_MAX_CSIZE && h[1] != 0 && h[1] <= cap && h[3] != 0 && h[3] <= csize
            && e0[0] != 0 && (e0[1] | e0[2] | e0[3]) == 0 && LZ4AMD_HINT_HEAD + ((uint64_t)e0[0] + 1) * LZ4AMD_HINT_ROW <= P.hint_stride) {
            hint = hp; total = h[1]; nseq = h[3]; nrows = e0[0];
        }
    }
    // ---- stage A: the record table (a malformed block ends here, nothing written).  On request it also writes the block's
    //      entry-point table, for the next decode of the same block.
    lz4amd_gdst make = nullptr;
    uint32_t make_rows = 0;
    // (lz4amd_k_decompress_runs: a run of ONE block may write its table too - kernels/chain_spec_kernel.h has the block decoded again, from the table)
    if (!hint && ok && !stored && (!chained || (kRuns && chain_run_head(P, b) && chain_run_tail(P, b))) && P.hint_make && P.hints && P.hint_stride >= 48) {
        make = LZ4AMD_TO_GDST((uint8_t*)P.hints + (uint64_t)b * P.hint_stride);
        make_rows = LZ4AMD_HINT_CAP_ROWS(P.hint_stride);
        if (tid == 0) *(uint32_t*)make = 0;                      // (no table until it is whole)
    }
    // (lz4amd_k_decompress_runs: a block whose twin - the same bytes against another history - was decoded by this workgroup a moment ago takes over
    //  what stage A found for it: the record table still lies in the workgroup's scratch)
    bool carried = false;
    uint32_t carried_minref = 0;
    if constexpr (kRuns) {
        if (twin_of != kNoTwin && ok && !stored) {
            const uint32_t* c = LZ4AMD_CHAIN_CARRY(P.chain, P.n_blocks) + 4u * twin_of;
            const uint32_t c0 = flag_load_agent(c);
            if (c0 != 0xFFFFFFFFu) {
                nseq = c0; total = flag_load_agent(c + 1); ridx = (uint32_t*)((char*)rectab + flag_load_agent(c + 2)); carried_minref = flag_load_agent(c + 3);
                carried = true;
                if (tid == 0) flag_store_agent(LZ4AMD_CHAIN_CARRY(P.chain, P.n_blocks) + 4u * b, LZ4AMD_CHAIN_CARRIED);      // (counted by lz4amd_plan_chain_stats)
            }
        }
    }
    if (!(kRuns && carried) && !hint && ok && !stored && !pre::preparse_block(src, csize, cap, prefix, rectab, smem, pre::table_bytes(csize), nseq, total, prof, &ridx, make, make_rows)) {
        if (!chained) { if (tid == 0) P.result[b] = err_at(((const uint32_t*)(smem + pre::kOffMisc))[pre::M_ERR]); return true; }
        ok = false;
    }
    if (prof && tid == 0) prof[1] = clock_ticks() - tstart;
    const uint32_t minref = (kRuns && carried) ? carried_minref : ((const uint32_t*)(smem + pre::kOffMisc))[pre::M_MINREF];
    if constexpr (kRuns) {
        if (!carried && ok && !stored && tid == 0) {                  // for the twin
            uint32_t* c = LZ4AMD_CHAIN_CARRY(P.chain, P.n_blocks) + 4u * b;
            flag_store_agent(c + 1, total); flag_store_agent(c + 2, (uint32_t)((const char*)ridx - (const char*)rectab)); flag_store_agent(c + 3, minref); flag_store_agent(c, nseq);
        }
    }
    __syncthreads();            // record table visible to the whole workgroup; stage A's LDS is dead

    long long start = 0;
    if (chained) {
        // -- where does my output start?  (published by the workgroup that owns block b-1)
        if (tid == 0) {
            long long s = 0;
            if (!chain_run_head(P, b)) while ((s = chain_load_acquire(&P.chain[b])) == -1) chain_wait_pause();
            misc[M_CHI] = (uint32_t)(unsigned long long)s; misc[M_SPARE] = (uint32_t)((unsigned long long)s >> 32);
        }
        __syncthreads();
        start = (long long)((unsigned long long)misc[M_CHI] | ((unsigned long long)misc[M_SPARE] << 32));
        __syncthreads();                                       // (stage B re-initialises those words)
        const unsigned long long before = (unsigned long long)(start < 0 ? 0 : start) + (P.prefix ? (unsigned long long)(uint32_t)P.prefix[b] : 0ull);
        prefix = before < kBias ? (uint32_t)before : kBias;
        // a predecessor failed, this block is malformed, or a match reaches before the start of the history (lz4.c:2356)
        if (start < 0 || !ok || (!stored && minref < kBias - prefix)) {
            if (tid == 0) { P.result[b] = -1; if (kRuns) flag_store_agent(&LZ4AMD_CHAIN_LOWREF(P.chain, P.n_blocks)[b], 0u); }
            if (!chain_run_tail(P, b)) chain_publish(P, b, -2); // the chain ends here
            return true;
        }
        dst = LZ4AMD_TO_GDST(P.dst[b]) + start;
        // (kernels/chain_spec_kernel.h wants to know: does a match read one of the first 256 bytes of the 64 KB in front of the run?)
        if (kRuns && tid == 0) flag_store_agent(&LZ4AMD_CHAIN_LOWREF(P.chain, P.n_blocks)[b], (!stored && (unsigned long long)minref + (unsigned long long)start < (unsigned long long)kBias - 65536u + 256u) ? 1u : 0u);
        if (stored) for (uint32_t i = tid; i < total; i += kDecThreads) dst[i] = src[i];
    }

    if (!stored && !stream_block(src, csize, dst, cap, prefix, rectab, ridx, nseq, total, smem, prof, hint, nrows)) {
        if (tid == 0 && P.hint_stats) atomicAdd(&P.hint_stats[1], 1u);
        return false;
    }
    if (hint && tid == 0 && P.hint_stats) atomicAdd(&P.hint_stats[0], 1u);
    if (make && tid == 0 && P.hint_stats) atomicAdd(&P.hint_stats[2], 1u);       // (a block that had more rows than room is counted too: its table stays invalid)

    __syncthreads();
    if (tid == 0) {
        P.result[b] = (int32_t)total;
        if (prof) prof[0] = clock_ticks() - tstart;
    }
    if (chained && !chain_run_tail(P, b)) chain_publish(P, b, start + (long long)total);
    return true;
}

// Workgroups pull blocks from a device-wide ticket counter (load balance for ragged batches).
constexpr uint32_t kGatedOut = 0xFFFFFFFEu;
template <bool kRuns>
__device__ __forceinline__ void decompress_batch_body_of(const DecBatch& P) {
    LZ4AMD_DYN_LDS(smem);
    uint32_t* misc = (uint32_t*)(smem + kOffMisc);
    uint32_t twin_next = 0, twin_src = kNoTwin;             // (kRuns) the block to decode next without a ticket: the twin of the one just decoded (lz4amd_dec_params.chain)
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            if (kRuns && twin_next) { misc[M_BLOCK] = twin_next - 1; misc[M_TWIN] = 0; }
            else {
                const uint32_t t = take_ticket(P.ticket);
                uint32_t b = (P.order && t < P.n_blocks) ? P.order[t] : t;
                if constexpr (kRuns) {
                    if (b < P.n_blocks) {                   // a gated block: is it wanted at all?  (the block that says so has a lower ticket: it is at work)
                        const uint32_t g = LZ4AMD_CHAIN_GATE(P.chain, P.n_blocks)[b];
                        if (g) {
                            uint32_t v;
                            while ((v = flag_load_agent(&LZ4AMD_CHAIN_LOWREF(P.chain, P.n_blocks)[g - 1])) == 0xFFFFFFFFu) chain_wait_pause();
                            if (v == 0) { P.result[b] = -1; b = kGatedOut; }
                        }
                    }
                    misc[M_TWIN] = b < P.n_blocks ? LZ4AMD_CHAIN_TWIN(P.chain, P.n_blocks)[b] : 0u;
                }
                misc[M_BLOCK] = b;
            }
        }
        __syncthreads();
        const uint32_t b = misc[M_BLOCK];
        if (kRuns && b == kGatedOut) continue;
        if (b >= P.n_blocks) break;
        uint32_t of = kNoTwin;
        if constexpr (kRuns) { if (twin_next) of = twin_src; twin_next = misc[M_TWIN]; twin_src = b; }
        bool use_hints = true;
        while (!decode_one_block<kRuns>(P, b, smem, use_hints, of)) { use_hints = false; __syncthreads(); }
    }
}
__device__ __forceinline__ void decompress_batch_body(const DecBatch& P) {      // (the interpreter's entry, tests/simt)
    if (P.chain) decompress_batch_body_of<true>(P); else decompress_batch_body_of<false>(P);
}

} // namespace lz4amd
