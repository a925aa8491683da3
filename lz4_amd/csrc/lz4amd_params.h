/* lz4amd_params.h -- kernel argument blocks shared by the C host code and the HIP kernels.
 * Plain C structs (pointers and sizes only); all pointers are DEVICE pointers. */
#ifndef LZ4AMD_PARAMS_H
#define LZ4AMD_PARAMS_H
#include <stdint.h>

typedef struct lz4amd_dec_params {
    const uint8_t* const* src;      /* [n] compressed blocks */
    const int32_t* src_size;        /* [n] */
    uint8_t* const* dst;            /* [n] output buffers */
    const int32_t* dst_cap;         /* [n] */
    const int32_t* prefix;          /* [n] or NULL: bytes of history right before dst (<= 64 KB used) */
    int32_t* result;                /* [n] decoded size, or negative on error */
    uint32_t n_blocks;
    uint32_t* ticket;               /* work-queue counter, zero before launch */
    uint8_t* scratch;               /* grid * scratch_stride bytes (per-workgroup segment tables) */
    uint64_t scratch_stride;
    uint64_t* prof;                 /* optional: 8 words per workgroup of phase timestamps */
    /* dependent blocks (lz4frame linked blocks): chain != NULL.  chain[i] = output bytes before block i (chain[0] = 0,
     * the others -1 before the launch, written by the workgroup that finishes block i-1; < -1: a predecessor failed).
     * Block i then writes at dst[i] + chain[i], sees min(64 KB, prefix[i] + chain[i]) bytes of history there, and
     * bit 0 of stored[i] marks a block that is copied as is (lz4frame.c:1758-1830).  Bit 1 marks the first block of a RUN: a launch may hold
     * several chains behind one another; a run's blocks share dst[] and prefix[], its first block starts at dst[i] and waits for nobody,
     * its last block publishes nothing (block 0 is a first block by itself). */
    long long* chain;               /* [n + 1] or NULL; behind the n + 1 words (LZ4AMD_CHAIN_BYTES: one allocation, the struct is what it was):
                                     *   lowref [n] uint32: every block reports 1 when one of its matches reads one of the first 256 bytes of the 64 KB in front
                                     *     of its run, else 0 (also when it fails); 0xFFFFFFFF = not known yet (set before the launch, like the words).
                                     *     kernels/chain_spec_kernel.h: two made-up histories tell all the other bytes apart
                                     *   gate [n] uint32, set when the plan is made: gate[b] = e + 1: block b is decoded only if block e reports lowref[e] != 0
                                     *     (the workgroup that drew b's ticket waits until e has reported: e's ticket must be lower); else its result is -1
                                     *     and it publishes nothing - all blocks of a run share their gate.  0 = no gate
                                     *   twin [n] uint32, set when the plan is made: twin[b] = t + 1: block t has the same compressed bytes as b (another made-up
                                     *     history, another destination) and NO ticket: the workgroup that decoded b decodes t right behind it, from b's
                                     *     record table (stage A once for the two); 0 = none
                                     *   carry [n] x 4 uint32 (between lowref and gate; reset with them): what stage A of block b found - { sequences, output bytes,
                                     *     where the region index lies in the workgroup's scratch, lowest position a match reads } - for b's twin */
    const uint8_t* stored;          /* [n] or NULL */
    const uint32_t* order;          /* [n] or NULL: the block the k-th ticket stands for (runs: first blocks of all runs, then second blocks, ... -
                                     * a workgroup per RUN is at work, and a block's predecessor still has the lower ticket) */
    /* entry-point tables ("hints", include/lz4amd.h): block i's table starts at hints + i * hint_stride; NULL = none.
     * Read only, never trusted: every entry is checked against the stream before a byte that depends on it is final. */
    const uint8_t* hints;
    uint64_t hint_stride;
    uint32_t* hint_stats;           /* optional: [0] += blocks decoded from their table, [1] += tables that did not fit their block, [2] += tables made */
    uint32_t hint_make;             /* != 0: a block whose table is missing or unusable is decoded the ordinary way AND gets its table written (the
                                     * hints memory must then be writable): the next decode of the same block parses from it */
} lz4amd_dec_params;

/* One block's entry-point table (layout: below, at LZ4AMD_HINT_MAGIC).  Row r names a sequence of the block's token chain: where its
 * token sits in the compressed block, where its literals start in the output, how many sequences precede it (mod 256: the decoder counts
 * them up from row to row).  Row 0 is the block's first sequence, row `nrows` the block's end; rows never decrease.  The
 * compressor writes about one row per 512 bytes of source and never fewer than one per 8 sequences - per 16 on data of fewer
 * than 32 bytes per sequence, whose rows would not fit the table's room otherwise (the distance, a power of two of sequences,
 * is set tile by tile from the tile's own number of sequences): every lane of the decoder's parser walks a row's
 * sequences one after the other, ~1000 cycles each, so rows must be short in sequences - and they must be short in bytes,
 * because the lanes' rows must lie in the 32 KB of the stream that are resident.  (Rows at fixed distances in the output -
 * an earlier layout - left two thirds of the lanes' steps idle: the sequences per KB vary threefold.)  Any table whose rows
 * lie on the chain works; one that does not is found out and costs time only. */
#define LZ4AMD_HINT_MAGIC 0x32485A4Cu           /* "LZH2": rows of 8 bytes (round 6; "LZ4H" had rows of 16) */
/* Layout: 32 bytes of header - { magic, out_size, csize, nseq }, { nrows, 0, 0, 0 } - then nrows + 1 rows of 8 bytes:
 * { token position (24 bits) | sequences before the row mod 256 (8 bits), output position }.  Row 0 is the block's first sequence { 0, 0 }, row nrows
 * the block's end { csize | nseq mod 256 << 24, out_size }.  Two neighbouring rows are at most 255 sequences apart (the decoder counts the sequences
 * before a row up from the differences), and a block whose compressed size does not fit 24 bits has no table. */
#define LZ4AMD_HINT_HEAD 32u
#define LZ4AMD_HINT_ROW 8u
#define LZ4AMD_HINT_MAX_CSIZE (1u << 24)
#define LZ4AMD_HINT_CAP_ROWS(stride) ((stride) >= 48 ? (uint32_t)(((stride) - LZ4AMD_HINT_HEAD) / LZ4AMD_HINT_ROW - 1) : 0u)      /* rows 0 .. cap - 1, and the end row */
#define LZ4AMD_HINT_EVERY_MAX 16u            /* sequences between two rows of a table lz4amd_k_compress writes, at most (data of fewer than 32 bytes
                                              * per sequence; 8 and fewer otherwise: about a row per 512 bytes) */
#define LZ4AMD_HINT_EVERY_LOG2 4u
#define LZ4AMD_CHAIN_BYTES(n) (((size_t)(n) + 1u) * 8u + (size_t)(n) * (4u + 16u + 4u + 4u))
#define LZ4AMD_CHAIN_RESET_BYTES(n) (((size_t)(n) + 1u) * 8u + (size_t)(n) * (4u + 16u))      /* what a launch sets to "not known yet": words, lowref, carry */
#define LZ4AMD_CHAIN_LOWREF(chain, n) ((uint32_t*)((long long*)(chain) + (size_t)(n) + 1u))
#define LZ4AMD_CHAIN_CARRY(chain, n) (LZ4AMD_CHAIN_LOWREF(chain, n) + (size_t)(n))
#define LZ4AMD_CHAIN_GATE(chain, n) (LZ4AMD_CHAIN_CARRY(chain, n) + 4u * (size_t)(n))
#define LZ4AMD_CHAIN_CARRIED 0xFFFFFFFDu       /* carry[b][0] of a block that was decoded from its twin's record table */
#define LZ4AMD_CHAIN_TWIN(chain, n) (LZ4AMD_CHAIN_GATE(chain, n) + (size_t)(n))
typedef struct lz4amd_hint_entry { uint32_t tok_ord, out; } lz4amd_hint_entry;

/* Linked blocks decoded side by side (kernels/chain_spec_kernel.h).  The chain is cut in UNITS of `group` consecutive blocks (the last one may be
 * shorter).  Unit 0 is decoded in place; unit u >= 1 against made-up 64 KB histories A, B (and C, if it has to be), into
 * slots[(3 * (u - 1) + v) * slot_stride + 65536] - each copy a run of dependent blocks of ONE launch of lz4amd_k_decompress (lz4amd_dec_params.chain).
 * spec_result: the results of that launch - entries 0 .. len(0) - 1: unit 0's blocks; len(0) + (u - 1) * 3 * group + v * len(u) + j: block j of
 * unit u, variant v = 0 (A), 1 (B), 2 (C: gated on what A's first block reports, lz4amd_dec_params.chain - its blocks answer -1 when the unit
 * does not need it); lowref: that launch's lowref words. */
typedef struct lz4amd_spec_params {
    uint32_t n, prefix0;            /* blocks of the chain; bytes of data right before out (<= 64 KB used) */
    uint32_t group, n_units;
    uint8_t* out;                   /* where block 0 starts; unit u at out + start[u] */
    uint8_t* slots; uint64_t slot_stride;
    const int32_t* spec_result;
    /* chains of large blocks (group = 1): variants B and C are decoded by a SECOND launch - independent blocks with 64 KB in front, from the entry-point
     * tables variant A's decode wrote - whose entry 2 (u - 1) is unit u's B and 2 (u - 1) + 1 its C.  spec_gate (between the launches) copies A's table
     * to both and sets C's source size to 0 where the unit does not need it.  All NULL / 0 otherwise. */
    const int32_t* spec_result_b;   /* [2 (n_units - 1)] results of the second launch */
    const uint8_t* tables_a;        /* the first launch's tables: unit u's (written by A's decode) at tables_a + (len(0) + 3 (u - 1)) * table_stride */
    uint8_t* tables_b;              /* the second launch's: entry e at tables_b + e * table_stride */
    uint64_t table_stride;
    const int32_t* src_size;        /* [n] compressed sizes of the chain's blocks */
    int32_t* b_src_size;            /* [2 (n_units - 1)] the second launch's source sizes */
    const uint32_t* lowref;
    long long* start;               /* [n_units] output bytes before unit u */
    int32_t* size;                  /* [n_units] decoded bytes of the unit's blocks up to its first bad one */
    int32_t* lastdep;               /* [n_units] last byte of the unit that is a copy of a history byte, -1: none */
    int32_t* badpos;                /* [n_units] first byte of the unit that refers to data before the start (lz4.c:2356), INT32_MAX: none */
    uint8_t* three;                 /* [n_units] != 0: the unit needs variant C */
    long long* done;                /* [n_units] != 0: the unit's bytes are final */
    uint32_t* info;                 /* [0] units to put together (the first one with a bad block is the last), [1] ticket counter */
    int32_t* result;                /* [n] decoded size per block, -1 from the first failure on */
} lz4amd_spec_params;

typedef struct lz4amd_comp_params {
    const uint8_t* const* src;      /* [n_blocks] */
    const int32_t* src_size;
    uint8_t* const* dst;
    const int32_t* dst_cap;
    const int32_t* prefix;          /* [n_blocks] or NULL: bytes of history right before src (<= 64 KB used) */
    int32_t* result;                /* [n_blocks] compressed size, 0 = failure */
    uint32_t n_blocks;
    uint32_t* ticket;               /* work-queue counter, zero before launch */
    uint64_t* prof;                 /* optional: 8 words per workgroup of phase cycle counts */
    uint8_t* hints;                 /* optional: block i's entry-point table is written at hints + i * hint_stride (layout below lz4amd_dec_params) */
    uint64_t hint_stride;
    int32_t acceleration;           /* LZ4_compress_fast's speed / ratio knob (lz4.c:1382-1400); values < 1 mean 1 */
} lz4amd_comp_params;

typedef struct lz4amd_hc_params {
    const uint8_t* const* src;      /* [n_blocks] */
    const int32_t* src_size;
    uint8_t* const* dst;
    const int32_t* dst_cap;
    int32_t* result;                /* [n_blocks] compressed size, 0 = failure */
    uint32_t n_blocks;
    uint32_t* ticket;               /* work-queue counter, zero before launch */
    uint8_t* scratch;               /* grid * scratch_stride bytes: per-workgroup chain / search state / sequence records */
    uint64_t scratch_stride;
    uint32_t max_src;               /* largest src_size of the table (fixes the scratch layout) */
    int32_t level;                  /* LZ4_compress_HC compressionLevel (lz4hc.h:66) */
    const int32_t* prefix;          /* [n_blocks] or NULL: bytes of history right before src (<= 64 KB used, rounded down to 64) */
    uint64_t* prof;                 /* optional: 8 words per workgroup of phase cycle counts */
    uint8_t* hints;                 /* optional: block i's entry-point table is written at hints + i * hint_stride (as by lz4amd_comp_params) */
    uint64_t hint_stride;
} lz4amd_hc_params;

typedef struct lz4amd_xxh_params {
    const uint8_t* const* src;      /* [n_blocks] */
    const int32_t* src_size;
    int32_t* result;                /* [n_blocks] XXH32(seed 0) of the block, as int32 */
    uint32_t n_blocks;
} lz4amd_xxh_params;

typedef struct lz4amd_gather_params {
    const uint8_t* const* src;      /* [n_blocks] */
    const int32_t* src_size;        /* [n_blocks] bytes to copy */
    uint8_t* const* dst;            /* [n_blocks] any alignment */
    const int32_t* dst_cap;
    int32_t* result;                /* [n_blocks] bytes copied, -1 if the row does not fit */
    uint32_t n_blocks;
} lz4amd_gather_params;

#endif
