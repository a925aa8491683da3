/*
 * lz4amd.h -- batch (block-table) entry points of the MI355X-native LZ4 block codec.
 *
 * This is the shape in which the reference itself drives the block codec when it has many
 * independent blocks: programs/bench.c:347-355 (`blockParam_t` = {srcPtr, srcSize, cPtr, cRoom,
 * cSize, resPtr, resSize}) looped over LZ4_compress_fast / LZ4_decompress_safe_usingDict
 * (bench.c:466-480, 522-542), and programs/lz4io.c:1130-1160 (one job per 4 MB chunk).
 * Here the whole table is handed to the GPU at once; every block is an independent LZ4 block
 * with exactly the semantics of
 *     LZ4_compress_default   (lib/lz4.h:191)   -> result = compressed size, 0 = failure
 *     LZ4_decompress_safe    (lib/lz4.h:208)   -> result = decoded size, < 0 = malformed / too small
 * Plain C ABI: pointers and sizes only.  Buffers named d_* are DEVICE (HBM) pointers; everything
 * else is host memory.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *
 * There is no CPU fallback: every call fails (LZ4AMD_E_NODEVICE) when no HIP device is usable.
 */
#ifndef LZ4AMD_H
#define LZ4AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LZ4AMD_OK            0
#define LZ4AMD_E_NODEVICE  (-1)   /* no usable HIP device / runtime error at init */
#define LZ4AMD_E_ARG       (-2)
#define LZ4AMD_E_MEMORY    (-3)
#define LZ4AMD_E_RUNTIME   (-4)   /* a HIP call failed; see lz4amd_last_error() */

typedef struct lz4amd_ctx  lz4amd_ctx;    /* per-device context */
typedef struct lz4amd_plan lz4amd_plan;   /* one block table bound to device buffers */

typedef enum {
    LZ4AMD_OP_COMPRESS   = 0,   /* LZ4_compress_default per block (lz4.c:1472) */
    LZ4AMD_OP_DECOMPRESS = 1,   /* LZ4_decompress_safe per block (lz4.c:2451) */
    LZ4AMD_OP_COMPRESS_HC = 2,  /* LZ4_compress_HC per block (lz4hc.c:1519) */
    LZ4AMD_OP_XXH32 = 3,        /* XXH32(seed 0) per block (xxhash.c:392); result = hash as int */
    LZ4AMD_OP_GATHER = 4        /* copy src_sizes[i] bytes of d_src[i] to d_dst[i], any alignment: packs compressed blocks
                                 * behind one another the way LZ4F_makeBlock appends them (lz4frame.c:883-914) */
} lz4amd_op;

int         lz4amd_ctx_create(lz4amd_ctx** out, int device);
void        lz4amd_ctx_destroy(lz4amd_ctx* ctx);
const char* lz4amd_last_error(void);
/* Arguments the reference acts on and this library accepts without acting on them are not errors, but they are not silent
 * either: the call records a notice (thread local, like the error text; "" when the last such call had nothing to say).
 * Today: LZ4_compress_fast* with acceleration > 2 (lz4.c:1389: the GPU parse knows two settings, 1 and 2) and
 * LZ4_compress_HC* with compressionLevel > 10 (lz4hc.c:92-106: levels 10 / 11 / 12 search 96 / 512 / 2048 candidates per position -
 * the reference's level 12: 16384 - and all three use level 10's 64-byte sufficient length). */
const char* lz4amd_last_notice(void);
int         lz4amd_device_cus(const lz4amd_ctx* ctx);

/* LZ4_compressBound (lz4.h:226) - pure arithmetic, usable without a device */
int         lz4amd_compress_bound(int src_size);

/* Bind a table of n blocks.  d_src[i]/d_dst[i] are device pointers, src_sizes/dst_caps host
 * arrays (copied).  level is used by LZ4AMD_OP_COMPRESS_HC only.
 * Device memory a plan holds besides the table itself: LZ4AMD_OP_COMPRESS none; LZ4AMD_OP_DECOMPRESS ~1/3 of the largest
 * compressed block per workgroup (one workgroup per CU at most; none of it is touched for blocks that come with an entry-point
 * table); LZ4AMD_OP_COMPRESS_HC ~18 bytes per byte of the largest source block per workgroup (chain, search state, parked walks,
 * records) - the number of workgroups is cut down so that this stays within 16 GiB (environment LZ4AMD_HC_SCRATCH_MB overrides):
 * 256 blocks of 4 MiB are compressed by ~220 workgroups, the rest of the table queues. */
int  lz4amd_plan_create(lz4amd_ctx* ctx, lz4amd_plan** out, lz4amd_op op, int n,
                        const void* const* d_src, const int* src_sizes,
                        void* const* d_dst, const int* dst_caps, int level);
/* LZ4AMD_OP_DECOMPRESS with history: block i may copy from the prefix_sizes[i] bytes that sit in
 * device memory right before d_dst[i] (LZ4_decompress_safe_usingDict in prefix mode, lz4.c:2719-2732
 * -> 2479 / 2504; this is how lz4frame decodes linked blocks, lz4frame.c:1901-1915).  At most the
 * last 64 KB are used.  The caller orders the launches so that the history is final. */
int  lz4amd_plan_create_prefix(lz4amd_ctx* ctx, lz4amd_plan** out, int n,
                               const void* const* d_src, const int* src_sizes,
                               void* const* d_dst, const int* dst_caps, const int* prefix_sizes);
/* LZ4AMD_OP_COMPRESS with history: block i may reference the prefix_sizes[i] bytes of source that
 * sit right before d_src[i] (linked blocks: lz4io.c:741-744, lz4frame.c:917-943 with
 * LZ4F_blockLinked; the reference does it with LZ4_compress_fast_continue, lz4.c:1707).  The
 * history is source data, so all blocks of a linked frame still compress in ONE launch.  The
 * largest multiple of 8 KB up to 64 KB is used. */
int  lz4amd_plan_create_compress_prefix(lz4amd_ctx* ctx, lz4amd_plan** out, int n,
                                        const void* const* d_src, const int* src_sizes,
                                        void* const* d_dst, const int* dst_caps, const int* prefix_sizes);
/* LZ4AMD_OP_COMPRESS_HC with history (LZ4_compress_HC_continue in prefix mode, lz4hc.c:1666-1700; lz4frame linked
 * blocks at HC levels): the largest multiple of 64 bytes up to 64 KB of prefix_sizes[i] is used. */
int  lz4amd_plan_create_compress_hc_prefix(lz4amd_ctx* ctx, lz4amd_plan** out, int n,
                                           const void* const* d_src, const int* src_sizes,
                                           void* const* d_dst, const int* dst_caps, const int* prefix_sizes, int level);
/* LZ4AMD_OP_DECOMPRESS over DEPENDENT blocks, all in one launch (lz4frame linked blocks, lz4frame.c:1901-1915:
 * LZ4_decompress_safe_usingDict with the previous 64 KB of output as dictionary, block after block).  Block i's output
 * starts where block i-1's ended: the packed output begins at d_dst0, where initial_prefix bytes of history (<= 64 KB
 * used) already sit in front of it; dst_caps[i] bounds block i's decoded size (the frame's maximum block size);
 * stored[i] != 0 marks a block that is copied as is (lz4frame.c:1758-1830), NULL = none.  results[i] = decoded size;
 * a malformed block and every block behind it report a negative value.
 * How (round 6, csrc/kernels/chain_spec_kernel.h): the chain is cut in units of 1 MiB of blocks; every unit but the first is decoded twice
 * against made-up histories (a third time where those two cannot tell: lz4amd_plan_chain_stats), side by side, and the bytes that come from the
 * real history are put in afterwards - a launch of the decoder, two bandwidth passes.  The plan then owns 3 x (64 KB + a unit) of device memory
 * per unit (3.2 GB for a GiB of 4 MiB blocks; budget 64 GiB, or LZ4AMD_CHAIN_SLOTS_MB); a plan that cannot have it, a chain of one unit, or
 * LZ4AMD_CHAIN_SERIAL=1 in the environment when the plan is made, decodes the blocks' copy stages one after the other (one CU at a time,
 * ~3.5 GB/s).  LZ4AMD_CHAIN_GROUP (blocks per unit), LZ4AMD_CHAIN_TWINS (0 / 1: a block's two copies decoded by one workgroup from one record
 * table) and LZ4AMD_CHAIN_TABLES (0 / 1, blocks of 1 MiB and more: the second copies by a second launch, from entry-point tables the first copies'
 * decode writes) override the plan's choices. */
int  lz4amd_plan_create_decompress_chained(lz4amd_ctx* ctx, lz4amd_plan** out, int n,
                                           const void* const* d_src, const int* src_sizes,
                                           void* d_dst0, const int* dst_caps, const unsigned char* stored, int initial_prefix);
/* a side-by-side plan of dependent blocks, after a launch: out = { units of the chain, units that were decoded a third time (a match in their
 * first 255 bytes reads one of the first 256 bytes of the 64 KB before them: two made-up histories cannot name those), bytes of units 1..
 * up to their unit's last byte that is a copy of a history byte (what the patch pass walks), decoded bytes of units 1.. }.  LZ4AMD_E_ARG for
 * any other plan; synchronises the device */
int  lz4amd_plan_chain_stats(lz4amd_plan* plan, unsigned long long out[4]);
/* Entry-point tables ("hints") - an optional, out-of-band column of the block table.
 * A compress plan (LZ4AMD_OP_COMPRESS) that has them attached writes, next to every block, a small table that names a
 * sequence of the block's token chain about every 512 bytes of source (every 2nd to 16th sequence): {position of its token in the block, position of its literals in the source,
 * sequences before it, mod 256} (8 bytes a row: 2.4 % of the source, 4.9 % of the compressed bytes of a `datagen -P60` block; blocks whose compressed
 * size needs more than 24 bits get none; layout: csrc/lz4amd_params.h).  The block itself is an ordinary
 * LZ4 block, byte for byte what it is without the table.  A decompress plan (LZ4AMD_OP_DECOMPRESS, lz4amd_plan_create /
 * _prefix) that has the tables attached parses every block from all its entries at once instead of first discovering the
 * serial token chain (what LZ4_decompress_generic's loop does implicitly, lz4.c:2123-2445) - about a third of the decoder's
 * time on blocks of a few MB.  The tables are never trusted: every entry is checked against the stream, with the same
 * rules as without them, before anything that depends on it becomes visible; a table that does not fit its block (wrong
 * block, stale, corrupt) only costs time - the block is then decoded without it, with the same result and error codes.
 * Block i's table lives at d_hints + i * stride (device memory, 16-byte aligned, stride a multiple of 16 and at least
 * lz4amd_hint_bytes(largest source / decoded size): room for a row per 128 bytes - a block that averages fewer than 16
 * bytes per sequence gets no table and is decoded without); it must stay valid while the plan is launched.
 * LZ4_decompress_safe, the frame API and every plan without tables are unaffected. */
size_t lz4amd_hint_bytes(int src_size);
int  lz4amd_plan_attach_hints(lz4amd_plan* plan, void* d_hints, size_t stride);
/* a decompress plan's count, since the tables were attached, of blocks decoded from their table and of tables that were
 * rejected (those blocks were decoded without); synchronises the device */
int  lz4amd_plan_hint_stats(lz4amd_plan* plan, unsigned* used, unsigned* rejected);
/* Tables for blocks of foreign origin (reference-compressed, out of a frame ...): with `on`, a decompress plan that has tables
 * attached writes the table of every block whose table is missing or unusable while it decodes the block the ordinary way (the
 * decoder's first stage knows every token then); the next launch - or any later plan over the same blocks and tables - parses
 * from it.  The hints memory must be writable.  lz4amd_plan_hints_made: tables written since they were attached. */
int  lz4amd_plan_make_hints(lz4amd_plan* plan, int on);
int  lz4amd_plan_hints_made(lz4amd_plan* plan, unsigned* made);
/* LZ4_compress_fast's `acceleration` (lz4.h:236, lz4.c:1382-1400) for the blocks of a LZ4AMD_OP_COMPRESS plan: 1 = default
 * (every second position of a block of 64 KB or more is probed), 2 and above = every fourth position: larger output, faster on
 * highly compressible data; clamped to 65537 like the reference (lz4.c:1386-1387). */
int  lz4amd_plan_set_acceleration(lz4amd_plan* plan, int acceleration);
void lz4amd_plan_destroy(lz4amd_plan* plan);
/* enqueue the whole table on `stream` (asynchronous) */
int  lz4amd_plan_launch(lz4amd_plan* plan, void* stream);
/* same, bracketing every kernel with HIP events on `stream`; blocks until done.
 * kernel_ms[0..3] receive the durations of the (up to 4) kernels of the op, total_ms their span. */
int  lz4amd_plan_launch_timed(lz4amd_plan* plan, void* stream, float kernel_ms[4], float* total_ms);
/* per-block results: device-resident int[n], or copied to the host (synchronises `stream`) */
const int* lz4amd_plan_device_results(const lz4amd_plan* plan);
int  lz4amd_plan_results(lz4amd_plan* plan, int* results, void* stream);

/* developer aid (set LZ4AMD_PROF=1 before plan_create): per-workgroup phase cycle stamps of the
 * decoder, 8 words per workgroup; returns the number of words copied */
int  lz4amd_plan_profile(lz4amd_plan* plan, unsigned long long* words, int max_words);

/* one-shot conveniences: plan + launch + results (synchronous) */
int  lz4amd_compress_batch(lz4amd_ctx* ctx, const void* const* d_src, const int* src_sizes,
                           void* const* d_dst, const int* dst_caps, int* results, int n, void* stream);
/* The `level` of the LZ4AMD_OP_COMPRESS_HC entry points: the reference's compressionLevel, optionally with this flag on top: the
 * optimal parse of levels 10-12 then prefers what decodes fast (no offsets below 8, lengths 19..36 cut to 18), like
 * LZ4_favorDecompressionSpeed (lz4hc.h:364; lz4hc.c:926-929, 1816-1818).  Ignored below level 10, as in the reference. */
#define LZ4AMD_HC_FAVOR_DEC_SPEED 0x100

/* LZ4_compress_HC (lz4hc.h:66) per block; level as in the reference (lz4hc.c:92-106): 1..2 = the two-table search (LZ4MID),
 * 3..9 = hash chain with 4..256 attempts, 10..12 = optimal parse; 0 and below = the default 9, above 12 = 12 */
int  lz4amd_compress_hc_batch(lz4amd_ctx* ctx, const void* const* d_src, const int* src_sizes,
                              void* const* d_dst, const int* dst_caps, int* results, int n, int level, void* stream);
int  lz4amd_decompress_batch(lz4amd_ctx* ctx, const void* const* d_src, const int* src_sizes,
                             void* const* d_dst, const int* dst_caps, int* results, int n, void* stream);

/* raw device-memory helpers for C callers that do not link the HIP runtime themselves */
void* lz4amd_dev_malloc(size_t bytes);
void  lz4amd_dev_free(void* d_ptr);
int   lz4amd_dev_upload(void* d_dst, const void* h_src, size_t bytes);
int   lz4amd_dev_download(void* h_dst, const void* d_src, size_t bytes);

/* Calibration for roofline reports: time `reps` launches of a plain 16-bytes-per-lane device copy of `bytes` bytes
 * (d_src -> d_dst, both in HBM) with HIP events on `stream`; *best_ms = fastest launch.  Bandwidth = 2 * bytes / time
 * (every byte is read once and written once). */
int  lz4amd_stream_copy_ms(lz4amd_ctx* ctx, void* d_dst, const void* d_src, size_t bytes, int reps, void* stream, float* best_ms);

#ifdef __cplusplus
}
#endif
#endif
