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
     * Block i then writes at dst[0] + chain[i], sees min(64 KB, prefix[0] + chain[i]) bytes of history there, and
     * stored[i] != 0 marks a block that is copied as is (lz4frame.c:1758-1830). */
    long long* chain;               /* [n + 1] or NULL */
    const uint8_t* stored;          /* [n] or NULL */
} lz4amd_dec_params;

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
