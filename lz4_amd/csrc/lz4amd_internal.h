/* lz4amd_internal.h -- private structures of the C host code. */
#ifndef LZ4AMD_INTERNAL_H
#define LZ4AMD_INTERNAL_H
#include "../../include/lz4amd.h"
#include "lz4amd_params.h"

#define LZ4AMD_PLAN_MAX_BUFS 20

struct lz4amd_ctx {
    int device;
    int n_cus;
};

struct lz4amd_plan {
    lz4amd_ctx* ctx;
    lz4amd_op op;
    int n;
    int level;
    unsigned grid;
    int* d_results;
    void* bufs[LZ4AMD_PLAN_MAX_BUFS];   /* every device allocation owned by the plan */
    void* ev[5];                        /* HIP events for the timed launch */
    lz4amd_dec_params dec;
    lz4amd_comp_params comp;
    lz4amd_hc_params hc;
    lz4amd_xxh_params xxh;
    lz4amd_gather_params gather;
    lz4amd_spec_params spec;            /* linked blocks decoded side by side: */
    struct lz4amd_plan* inner;          /* ... the launch of dependent blocks in runs that does the decoding (owned) */
    struct lz4amd_plan* inner_b;        /* ... chains of large blocks: the second copies, decoded from the tables the first ones' decode wrote (owned; NULL: they are part of `inner`) */
    unsigned spec_max_cap;
    int row0[2];                        /* lz4amd_plan_set_row0: host copy of the sizes in flight */
};

void lz4amd_set_error(const char* msg);
void lz4amd_set_notice(const char* msg);     /* an argument that was accepted and not acted on (include/lz4amd.h) */

/* a one-block plan kept by a thread: new sizes (and HC level) for its only row, sent on `stream` before the launch */
int lz4amd_plan_set_row0(lz4amd_plan* p, int src_size, int dst_cap, int level, void* stream);
/* ... or its sizes and its result live in page-locked host memory the device reads and writes directly
 * (row[0] = source size, row[1] = capacity, row[2] = result): no copy calls for them at all */
int lz4amd_plan_bind_host_row(lz4amd_plan* p, int* row);
void lz4amd_plan_set_level(lz4amd_plan* p, int level);      /* LZ4_compress_HC level of the next launch */

/* process-wide default context used by the classic one-block API (lz4_api.c) */
lz4amd_ctx* lz4amd_default_ctx(void);

/* one host block through the device (lz4_api.c): upload, one-row plan, download; `fail` is returned
 * when the device path cannot run */
int lz4amd_compress_with_history(const char* hist, int histSize, const char* src, char* dst, int srcSize, int dstCapacity, int hc_level);
int lz4amd_run_one(lz4amd_op op, const char* src, char* dst, int srcSize, int dstCapacity, int level, int fail);

#endif
