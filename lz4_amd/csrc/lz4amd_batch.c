/*
 * lz4amd_batch.c -- host side (plain C) of the batch block codec: contexts, block-table plans,
 * launch sequencing.  All GPU work goes through the thin FFI of lz4amd_ffi.h.
 *
 * A "plan" is the device-resident image of the reference's block table
 * (programs/bench.c:347-355 blockParam_t): per-block source/destination pointers and sizes,
 * per-block results, and the scratch the kernels need (sequence tables for the decoder,
 * match records / sub-chunk tables for the compressor).
 */
#include "../../include/lz4amd.h"
#include "lz4amd_ffi.h"
#include "lz4amd_internal.h"
#include "lz4amd_params.h"
#define LZ4AMD_TRACE_BYTES (4u << 20)
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

static __thread char g_last_error[256] = "";

void lz4amd_set_error(const char* msg)
{
    snprintf(g_last_error, sizeof g_last_error, "%s", msg ? msg : "");
}
const char* lz4amd_last_error(void) { return g_last_error; }
static __thread char g_last_notice[160] = "";
void lz4amd_set_notice(const char* msg) { snprintf(g_last_notice, sizeof g_last_notice, "%s", msg ? msg : ""); }
const char* lz4amd_last_notice(void) { return g_last_notice; }

int lz4amd_compress_bound(int n)
{   /* lz4.h:214-215 LZ4_COMPRESSBOUND */
    if (n < 0 || (unsigned)n > 0x7E000000u) return 0;
    return n + n / 255 + 16;
}

/* ------------------------------------------------------------------ context */
int lz4amd_ctx_create(lz4amd_ctx** out, int device)
{
    lz4amd_ctx* c;
    int cus = 0;
    if (!out) return LZ4AMD_E_ARG;
    *out = NULL;
    if (lz4amd_hip_init(device, &cus) != 0) {
        lz4amd_set_error(lz4amd_hip_errstr());
        fprintf(stderr, "lz4_amd: cannot use HIP device %d: %s (this library has no CPU fallback)\n",
                device, lz4amd_hip_errstr());
        return LZ4AMD_E_NODEVICE;
    }
    c = (lz4amd_ctx*)calloc(1, sizeof *c);
    if (!c) return LZ4AMD_E_MEMORY;
    c->device = device;
    c->n_cus = cus > 0 ? cus : 1;
    *out = c;
    return LZ4AMD_OK;
}

void lz4amd_ctx_destroy(lz4amd_ctx* ctx) { free(ctx); }
int lz4amd_device_cus(const lz4amd_ctx* ctx) { return ctx ? ctx->n_cus : 0; }

void* lz4amd_dev_malloc(size_t bytes) { return lz4amd_hip_malloc(bytes); }
void  lz4amd_dev_free(void* p) { lz4amd_hip_free(p); }
int   lz4amd_dev_upload(void* d, const void* h, size_t n)
{ if (lz4amd_hip_h2d(d, h, n, NULL) || lz4amd_hip_sync(NULL)) { lz4amd_set_error(lz4amd_hip_errstr()); return LZ4AMD_E_RUNTIME; } return 0; }
int   lz4amd_dev_download(void* h, const void* d, size_t n)
{ if (lz4amd_hip_d2h(h, d, n, NULL) || lz4amd_hip_sync(NULL)) { lz4amd_set_error(lz4amd_hip_errstr()); return LZ4AMD_E_RUNTIME; } return 0; }

/* LZ4_compress_HC's compressionLevel as the kernel takes it: the reference's clamp (lz4hc.c:1414-1415: below 1 the default 9,
 * above 12 level 12), LZ4AMD_HC_FAVOR_DEC_SPEED kept on top.  (The kernel looks at the low byte only: -1 must not become 255.) */
static int hc_level_norm(int level)
{
    int flag = 0;
    if (level >= LZ4AMD_HC_FAVOR_DEC_SPEED && level < 2 * LZ4AMD_HC_FAVOR_DEC_SPEED) { flag = LZ4AMD_HC_FAVOR_DEC_SPEED; level -= LZ4AMD_HC_FAVOR_DEC_SPEED; }
    if (level < 1) level = 9;
    if (level > 12) level = 12;
    return level | flag;
}

/* --------------------------------------------------------------------- plan */
static void* dev_array(const void* host, size_t bytes, int* err)
{
    void* d = lz4amd_hip_malloc(bytes ? bytes : 16);
    if (!d) { *err = LZ4AMD_E_MEMORY; return NULL; }
    if (host && bytes && lz4amd_hip_h2d(d, host, bytes, NULL)) *err = LZ4AMD_E_RUNTIME;
    return d;
}

void lz4amd_plan_destroy(lz4amd_plan* p)
{
    int i;
    if (!p) return;
    if (p->ctx) (void)lz4amd_hip_use_device(p->ctx->device);
    lz4amd_plan_destroy(p->inner); lz4amd_plan_destroy(p->inner_b);
    for (i = 0; i < LZ4AMD_PLAN_MAX_BUFS; i++) lz4amd_hip_free(p->bufs[i]);
    for (i = 0; i < 5; i++) lz4amd_hip_event_destroy(p->ev[i]);
    free(p);
}

int lz4amd_plan_create(lz4amd_ctx* ctx, lz4amd_plan** out, lz4amd_op op, int n,
                       const void* const* d_src, const int* src_sizes,
                       void* const* d_dst, const int* dst_caps, int level)
{
    lz4amd_plan* p;
    int err = 0, nb = 0, i;
    size_t un = (size_t)(n > 0 ? n : 0);
    if (!ctx || !out || n < 0 || (n > 0 && (!d_src || !src_sizes))) return LZ4AMD_E_ARG;
    if (n > 0 && op != LZ4AMD_OP_XXH32 && (!d_dst || !dst_caps)) return LZ4AMD_E_ARG;
    *out = NULL;
    if (lz4amd_hip_use_device(ctx->device)) { lz4amd_set_error(lz4amd_hip_errstr()); return LZ4AMD_E_RUNTIME; }
    p = (lz4amd_plan*)calloc(1, sizeof *p);
    if (!p) return LZ4AMD_E_MEMORY;
    p->ctx = ctx; p->op = op; p->n = n; p->level = level;

    /* the block table itself */
    void* dsrc  = p->bufs[nb++] = dev_array(d_src, un * sizeof(void*), &err);
    void* dssz  = p->bufs[nb++] = dev_array(src_sizes, un * sizeof(int), &err);
    void* ddst  = p->bufs[nb++] = dev_array(d_dst, un * sizeof(void*), &err);
    void* dcap  = p->bufs[nb++] = dev_array(dst_caps, un * sizeof(int), &err);
    void* dres  = p->bufs[nb++] = dev_array(NULL, un * sizeof(int), &err);
    p->d_results = (int*)dres;

    if (op == LZ4AMD_OP_DECOMPRESS) {
        unsigned max_c = 0, max_cap = 0, grid;
        lz4amd_dec_params* q = &p->dec;
        for (i = 0; i < n; i++) {
            if (src_sizes[i] > 0 && (unsigned)src_sizes[i] > max_c) max_c = (unsigned)src_sizes[i];
            if (dst_caps[i] > 0 && (unsigned)dst_caps[i] > max_cap) max_cap = (unsigned)dst_caps[i];
        }
        /* one 1024-thread workgroup owns a CU's LDS; blocks are pulled from a ticket counter */
        grid = (unsigned)ctx->n_cus;
        if ((unsigned)n < grid) grid = (unsigned)n;
        p->grid = grid;
        q->src = (const uint8_t* const*)dsrc; q->src_size = (const int32_t*)dssz;
        q->dst = (uint8_t* const*)ddst; q->dst_cap = (const int32_t*)dcap;
        q->result = (int32_t*)dres; q->n_blocks = (uint32_t)n;
        q->prefix = NULL; q->hints = NULL; q->hint_stride = 0; q->hint_stats = NULL; q->hint_make = 0;
        q->scratch_stride = (lz4amd_hip_dec_scratch_bytes(max_c, max_cap) + 255) & ~(uint64_t)255;
        q->prof = NULL;
        if (getenv("LZ4AMD_PROF")) {            /* developer aid: per-workgroup phase timestamps */
            /* (behind the workgroups' words: room for the event trace of a developer build, -DLZ4AMD_DEC_TRACE) */
            q->prof = (uint64_t*)(p->bufs[nb++] = dev_array(NULL, (size_t)(grid ? grid : 1) * 64 + LZ4AMD_TRACE_BYTES, &err));
            if (q->prof && lz4amd_hip_memset(q->prof, 0, (size_t)(grid ? grid : 1) * 64 + LZ4AMD_TRACE_BYTES, NULL)) err = LZ4AMD_E_RUNTIME;
        }
        q->ticket = (uint32_t*)(p->bufs[nb++] = dev_array(NULL, 64, &err));
        q->scratch = (uint8_t*)(p->bufs[nb++] = dev_array(NULL, (size_t)q->scratch_stride * (grid ? grid : 1), &err));
    } else if (op == LZ4AMD_OP_COMPRESS) {
        lz4amd_comp_params* q = &p->comp;
        /* one 1024-thread workgroup streams through a block (source ring + hash table in LDS);
         * blocks are pulled from a ticket counter */
        unsigned grid = (unsigned)ctx->n_cus;
        if ((unsigned)n < grid) grid = (unsigned)n;
        p->grid = grid;
        q->src = (const uint8_t* const*)dsrc; q->src_size = (const int32_t*)dssz;
        q->dst = (uint8_t* const*)ddst; q->dst_cap = (const int32_t*)dcap;
        q->result = (int32_t*)dres; q->n_blocks = (uint32_t)n;
        q->ticket = (uint32_t*)(p->bufs[nb++] = dev_array(NULL, 64, &err));
        q->prefix = NULL; q->hints = NULL; q->hint_stride = 0; q->acceleration = 1;
        q->prof = NULL;
        if (getenv("LZ4AMD_PROF")) {            /* developer aid: per-workgroup phase cycle counts */
            q->prof = (uint64_t*)(p->bufs[nb++] = dev_array(NULL, (size_t)(grid ? grid : 1) * 64, &err));
            if (q->prof && lz4amd_hip_memset(q->prof, 0, (size_t)(grid ? grid : 1) * 64, NULL)) err = LZ4AMD_E_RUNTIME;
        }
    } else if (op == LZ4AMD_OP_COMPRESS_HC) {
        lz4amd_hc_params* q = &p->hc;
        /* one 1024-thread workgroup per block at a time (head table / band rings in LDS); the chain,
         * the per-position search state and the sequence records of the block in flight live in a
         * per-workgroup scratch area sized for the largest block of the table */
        unsigned grid = (unsigned)ctx->n_cus, max_n = 0;
        for (i = 0; i < n; i++) if (src_sizes[i] > 0 && (unsigned)src_sizes[i] > max_n) max_n = (unsigned)src_sizes[i];
        if ((unsigned)n < grid) grid = (unsigned)n;
        p->grid = grid;
        q->src = (const uint8_t* const*)dsrc; q->src_size = (const int32_t*)dssz;
        q->dst = (uint8_t* const*)ddst; q->dst_cap = (const int32_t*)dcap;
        q->result = (int32_t*)dres; q->n_blocks = (uint32_t)n;
        q->level = hc_level_norm(level); q->max_src = max_n + 65536;           /* room for a block's history (lz4amd_plan_create_compress_hc_prefix) */
        q->prefix = NULL; q->hints = NULL; q->hint_stride = 0;
        q->scratch_stride = (lz4amd_hip_hc_scratch_bytes(q->max_src) + 255) & ~(uint64_t)255;
        /* ~18 bytes of scratch per source byte of the largest block and workgroup (include/lz4amd.h): a table of large blocks is
         * compressed by as many workgroups as fit a budget (16 GiB, or LZ4AMD_HC_SCRATCH_MB) - the blocks queue on them - and never by
         * fewer than one */
        {   const char* e = getenv("LZ4AMD_HC_SCRATCH_MB");
            unsigned long long budget = e && atoll(e) > 0 ? (unsigned long long)atoll(e) << 20 : (unsigned long long)16 << 30;
            unsigned long long fit = budget / (q->scratch_stride ? q->scratch_stride : 1);
            if (fit < 1) fit = 1;
            if (grid > fit) grid = (unsigned)fit;
            p->grid = grid;
        }
        q->prof = NULL;
        if (getenv("LZ4AMD_PROF")) {
            q->prof = (uint64_t*)(p->bufs[nb++] = dev_array(NULL, (size_t)(grid ? grid : 1) * 64, &err));
            if (q->prof && lz4amd_hip_memset(q->prof, 0, (size_t)(grid ? grid : 1) * 64, NULL)) err = LZ4AMD_E_RUNTIME;
        }
        q->ticket = (uint32_t*)(p->bufs[nb++] = dev_array(NULL, 64, &err));
        q->scratch = (uint8_t*)(p->bufs[nb++] = dev_array(NULL, (size_t)q->scratch_stride * (grid ? grid : 1), &err));
    } else if (op == LZ4AMD_OP_XXH32) {
        lz4amd_xxh_params* q = &p->xxh;          /* d_dst / dst_caps are ignored: the result is the hash */
        p->grid = (unsigned)n;
        q->src = (const uint8_t* const*)dsrc; q->src_size = (const int32_t*)dssz;
        q->result = (int32_t*)dres; q->n_blocks = (uint32_t)n;
    } else if (op == LZ4AMD_OP_GATHER) {
        lz4amd_gather_params* q = &p->gather;
        p->grid = (unsigned)n;
        q->src = (const uint8_t* const*)dsrc; q->src_size = (const int32_t*)dssz;
        q->dst = (uint8_t* const*)ddst; q->dst_cap = (const int32_t*)dcap;
        q->result = (int32_t*)dres; q->n_blocks = (uint32_t)n;
    } else {
        lz4amd_set_error("operation not implemented on the device yet");
        lz4amd_plan_destroy(p);
        return LZ4AMD_E_ARG;
    }
    if (!err && lz4amd_hip_sync(NULL)) err = LZ4AMD_E_RUNTIME;
    if (err) {
        lz4amd_set_error(err == LZ4AMD_E_MEMORY ? "device allocation failed" : lz4amd_hip_errstr());
        lz4amd_plan_destroy(p);
        return err;
    }
    *out = p;
    return LZ4AMD_OK;
}

static int plan_create_with_prefix(lz4amd_ctx* ctx, lz4amd_plan** out, lz4amd_op op, int n,
                                   const void* const* d_src, const int* src_sizes,
                                   void* const* d_dst, const int* dst_caps, const int* prefix_sizes, int level)
{
    int rc = lz4amd_plan_create(ctx, out, op, n, d_src, src_sizes, d_dst, dst_caps, level);
    int err = 0, i;
    if (rc || !prefix_sizes || n <= 0) return rc;
    for (i = 0; i < LZ4AMD_PLAN_MAX_BUFS && (*out)->bufs[i]; i++) {}
    if (i >= LZ4AMD_PLAN_MAX_BUFS) { lz4amd_plan_destroy(*out); *out = NULL; return LZ4AMD_E_MEMORY; }
    (*out)->bufs[i] = dev_array(prefix_sizes, (size_t)n * sizeof(int), &err);
    if (!err && lz4amd_hip_sync(NULL)) err = LZ4AMD_E_RUNTIME;
    if (err) { lz4amd_plan_destroy(*out); *out = NULL; return err; }
    if (op == LZ4AMD_OP_DECOMPRESS) (*out)->dec.prefix = (const int32_t*)(*out)->bufs[i];
    else if (op == LZ4AMD_OP_COMPRESS_HC) (*out)->hc.prefix = (const int32_t*)(*out)->bufs[i];
    else (*out)->comp.prefix = (const int32_t*)(*out)->bufs[i];
    return LZ4AMD_OK;
}
int lz4amd_plan_create_prefix(lz4amd_ctx* ctx, lz4amd_plan** out, int n,
                              const void* const* d_src, const int* src_sizes,
                              void* const* d_dst, const int* dst_caps, const int* prefix_sizes)
{ return plan_create_with_prefix(ctx, out, LZ4AMD_OP_DECOMPRESS, n, d_src, src_sizes, d_dst, dst_caps, prefix_sizes, 0); }
int lz4amd_plan_create_compress_prefix(lz4amd_ctx* ctx, lz4amd_plan** out, int n,
                                       const void* const* d_src, const int* src_sizes,
                                       void* const* d_dst, const int* dst_caps, const int* prefix_sizes)
{ return plan_create_with_prefix(ctx, out, LZ4AMD_OP_COMPRESS, n, d_src, src_sizes, d_dst, dst_caps, prefix_sizes, 0); }
int lz4amd_plan_create_compress_hc_prefix(lz4amd_ctx* ctx, lz4amd_plan** out, int n,
                                          const void* const* d_src, const int* src_sizes,
                                          void* const* d_dst, const int* dst_caps, const int* prefix_sizes, int level)
{ return plan_create_with_prefix(ctx, out, LZ4AMD_OP_COMPRESS_HC, n, d_src, src_sizes, d_dst, dst_caps, prefix_sizes, level); }

/* Dependent blocks in one launch, general form: block i writes behind the blocks of its run at dsts[i] (the same for all blocks of a run) with
 * prefixes[i] bytes of history in front of the run; flags[i]: bit 0 stored block, bit 1 first block of a run (lz4amd_dec_params.chain). */
static int chained_runs_create(lz4amd_ctx* ctx, lz4amd_plan** out, int n, const void* const* d_src, const int* src_sizes,
                               void* const* dsts, const int* dst_caps, const unsigned char* flags, const int* prefixes)
{
    int rc, err = 0, k;
    lz4amd_plan* p;
    rc = plan_create_with_prefix(ctx, out, LZ4AMD_OP_DECOMPRESS, n, d_src, src_sizes, dsts, dst_caps, prefixes, 0);
    if (rc) return rc;
    p = *out;
    for (k = 0; k < LZ4AMD_PLAN_MAX_BUFS && p->bufs[k]; k++) {}
    if (k + 2 > LZ4AMD_PLAN_MAX_BUFS) { lz4amd_plan_destroy(p); *out = NULL; return LZ4AMD_E_MEMORY; }
    p->dec.chain = (long long*)(p->bufs[k] = dev_array(NULL, LZ4AMD_CHAIN_BYTES(n), &err));          /* (chain words, lowref, gate: lz4amd_params.h) */
    if (!err && lz4amd_hip_memset(LZ4AMD_CHAIN_GATE(p->dec.chain, n), 0, (size_t)n * 2u * sizeof(uint32_t), NULL)) err = LZ4AMD_E_RUNTIME;      /* no block is gated, none has a twin */
    if (flags) p->dec.stored = (const uint8_t*)(p->bufs[k + 1] = dev_array(flags, (size_t)n, &err));
    if (!err && lz4amd_hip_sync(NULL)) err = LZ4AMD_E_RUNTIME;
    if (err) { lz4amd_plan_destroy(p); *out = NULL; return err; }
    return LZ4AMD_OK;
}

/* one more device array of a plan that was just made */
static void* plan_add_array(lz4amd_plan* p, const void* host, size_t bytes, int* err)
{
    int k;
    for (k = 0; k < LZ4AMD_PLAN_MAX_BUFS && p->bufs[k]; k++) {}
    if (k >= LZ4AMD_PLAN_MAX_BUFS) { *err = LZ4AMD_E_MEMORY; return NULL; }
    return p->bufs[k] = dev_array(host, bytes, err);
}

/* Linked blocks side by side (kernels/chain_spec_kernel.h): the chain is cut in units of `group` blocks; the plan owns ONE launch of dependent
 * blocks in runs - unit 0 in place, every other unit against made-up histories A, B and (gated: only if A's first block says so) C, in slots
 * of its own - and the tables of the passes that put the output together.  Costs 3 x (64 KB + a unit) of device memory per unit; a plan
 * that cannot have it, or whose chain is one unit, decodes one block after the other. */
static int chained_spec_create(lz4amd_ctx* ctx, lz4amd_plan** out, int n,
                               const void* const* d_src, const int* src_sizes,
                               void* d_dst0, const int* dst_caps, const unsigned char* stored, int initial_prefix)
{
    size_t un = (size_t)n, ne, nu, G, len0, e, stride;
    lz4amd_plan* p;
    const void** esrc = NULL; void** edst = NULL; int *esz = NULL, *ecap = NULL, *epre = NULL; unsigned char* efl = NULL; unsigned *ord = NULL, *gate = NULL, *twin = NULL;
    unsigned max_cap = 0;
    size_t nt_pad = 0;
    int err = 0, nb = 0, i, rc, twins, tables;
    size_t hstride = 0;
    lz4amd_spec_params* q;
    for (i = 0; i < n; i++) {
        if (dst_caps[i] <= 0) return LZ4AMD_E_ARG;
        if ((unsigned)dst_caps[i] > max_cap) max_cap = (unsigned)dst_caps[i];
    }
    {   /* blocks per unit: as many as make 1 MiB (a unit's blocks are decoded one after the other, units side by side: the dependent stretch at
         * a unit's start - ~100 KB on datagen -P60 - should be a small part of it) */
        const char* g = getenv("LZ4AMD_CHAIN_GROUP");
        G = g && atoi(g) > 0 ? (size_t)atoi(g) : ((size_t)1 << 20) / max_cap;
        if (G < 1) G = 1;
        if (G * (size_t)max_cap > ((size_t)1 << 30)) G = ((size_t)1 << 30) / max_cap;     /* (positions inside a unit are 31-bit numbers) */
        if (G < 1) return LZ4AMD_E_ARG;
    }
    nu = (un + G - 1) / G;
    if (nu < 2) return LZ4AMD_E_ARG;
    len0 = G < un ? G : un;
    ne = len0 + 3 * (un - len0);
    {   /* chains of large blocks (a unit = one block, none of them stored): A's decode writes the block's entry-point table, B is decoded by a second
         * launch from it, without the decoder's first stage (kernels/chain_spec_kernel.h) */
        const char* tb = getenv("LZ4AMD_CHAIN_TABLES");
        tables = tb ? atoi(tb) != 0 : 2 * (nu - 1) + 1 > (size_t)ctx->n_cus;      /* (a chain whose two copies per block find a CU each decodes them at once: 64 x 4 MiB 2.8 against 4.1 ms) */
        if (G != 1) tables = 0;
        for (i = (int)len0; tables && stored && i < n; i++) if (stored[i]) tables = 0;
        for (i = (int)len0; tables && i < n; i++) if ((unsigned)src_sizes[i] >= LZ4AMD_HINT_MAX_CSIZE) tables = 0;
    }
    {   /* twins (lz4amd_dec_params.chain): A's and B's copy of a block decoded by one workgroup, stage A once.  Measured: pays for blocks
         * of up to 512 KB when the runs outnumber the CUs (4096 x 64 KiB: 4.7 -> 3.7 ms); a 4 MiB block's second copy stage finds neither its records nor
         * its compressed bytes in the L2 any more and takes longer than a whole decode (3.87 -> 4.07 ms), and a short chain has CUs to spare */
        const char* tw = getenv("LZ4AMD_CHAIN_TWINS");
        twins = tw ? atoi(tw) != 0 : (G > 1 && 2 * (nu - 1) >= (size_t)ctx->n_cus);
        if (tables) twins = 0;
    }
    stride = (65536u + G * (size_t)max_cap + 64u + 255u) & ~(size_t)255u;
    {   const char* lim = getenv("LZ4AMD_CHAIN_SLOTS_MB");                     /* (the slots' budget: 64 GiB of the 288) */
        const unsigned long long budget = lim && atoll(lim) > 0 ? (unsigned long long)atoll(lim) << 20 : (unsigned long long)64 << 30;
        if ((unsigned long long)stride * 3ull * (nu - 1) > budget) return LZ4AMD_E_MEMORY;
    }
    if (lz4amd_hip_use_device(ctx->device)) { lz4amd_set_error(lz4amd_hip_errstr()); return LZ4AMD_E_RUNTIME; }
    p = (lz4amd_plan*)calloc(1, sizeof *p);
    if (!p) return LZ4AMD_E_MEMORY;
    p->ctx = ctx; p->op = LZ4AMD_OP_DECOMPRESS; p->n = n; p->spec_max_cap = (unsigned)(G * max_cap);
    q = &p->spec;
    q->n = (uint32_t)n; q->prefix0 = (uint32_t)initial_prefix; q->group = (uint32_t)G; q->n_units = (uint32_t)nu;
    q->out = (uint8_t*)d_dst0; q->slot_stride = stride;
    q->result = (int32_t*)(p->bufs[nb++] = dev_array(NULL, un * sizeof(int), &err));
    p->d_results = (int*)q->result;
    q->start = (long long*)(p->bufs[nb++] = dev_array(NULL, nu * sizeof(long long), &err));
    q->done = (long long*)(p->bufs[nb++] = dev_array(NULL, nu * sizeof(long long), &err));
    q->size = (int32_t*)(p->bufs[nb++] = dev_array(NULL, nu * sizeof(int), &err));
    q->lastdep = (int32_t*)(p->bufs[nb++] = dev_array(NULL, nu * sizeof(int), &err));
    q->badpos = (int32_t*)(p->bufs[nb++] = dev_array(NULL, nu * sizeof(int), &err));
    q->three = (uint8_t*)(p->bufs[nb++] = dev_array(NULL, nu, &err));
    q->info = (uint32_t*)(p->bufs[nb++] = dev_array(NULL, 64, &err));
    if (!err) q->slots = (uint8_t*)(p->bufs[nb++] = dev_array(NULL, stride * 3 * (nu - 1), &err));
    esrc = (const void**)malloc(ne * sizeof *esrc); edst = (void**)malloc(ne * sizeof *edst);
    esz = (int*)malloc(ne * sizeof *esz); ecap = (int*)malloc(ne * sizeof *ecap); epre = (int*)malloc(ne * sizeof *epre); efl = (unsigned char*)malloc(ne);
    ord = (unsigned*)malloc(ne * sizeof *ord); gate = (unsigned*)calloc(ne, sizeof *gate); twin = (unsigned*)calloc(ne, sizeof *twin);
    if (!esrc || !edst || !esz || !ecap || !epre || !efl || !ord || !gate || !twin) err = LZ4AMD_E_MEMORY;
    if (!err) {
        /* entries 0 .. len0 - 1: unit 0 where it belongs; then unit u, variant v, block j at len0 + (u - 1) * 3 * G + v * len(u) + j.
         * Tickets (lz4amd_dec_params.order): the first blocks of all runs - A's, then C's -, then the second ones ...: a workgroup per
         * RUN is at work, a block's predecessor and a gated block's gate have the lower ticket. */
        size_t u, v, j, t = 0;
        static const size_t by_ticket[3] = {0, 2, 1};
        for (j = 0; j < len0; j++) {
            esrc[j] = d_src[j]; edst[j] = d_dst0; esz[j] = src_sizes[j]; ecap[j] = dst_caps[j]; epre[j] = initial_prefix;
            efl[j] = (unsigned char)((stored && stored[j] ? 1 : 0) | (j == 0 ? 2 : 0));
        }
        e = len0;
        for (u = 1; u < nu; u++) {
            const size_t lenu = un - u * G < G ? un - u * G : G, a_head = len0 + (u - 1) * 3 * G;
            /* C is asked for by A's FIRST block only if no later block can begin within the unit's first 255 bytes: a block of c compressed
             * bytes decodes to c - c / 255 - 2 at least (a sequence of s bytes with k length bytes for its literals yields s + 1 - k; stored: c) */
            const int ask = lenu == 1 || src_sizes[u * G] >= 512;
            for (v = 0; v < 3; v++)
                for (j = 0; j < lenu; j++, e++) {
                    const size_t b = u * G + j;
                    esrc[e] = d_src[b]; edst[e] = q->slots + ((u - 1) * 3 + v) * stride + 65536; esz[e] = src_sizes[b]; ecap[e] = dst_caps[b]; epre[e] = 65536;
                    efl[e] = (unsigned char)((stored && stored[b] ? 1 : 0) | (j == 0 ? 2 : 0));
                    if (v == 2 && ask) gate[e] = (unsigned)(a_head + 1);
                }
        }
        for (j = 0; j < G; j++) {
            if (j < len0) ord[t++] = (unsigned)j;
            for (v = 0; v < (tables ? 1u : twins ? 2u : 3u); v++)
                for (u = 1; u < nu; u++) {
                    const size_t lenu = un - u * G < G ? un - u * G : G;
                    if (j < lenu) ord[t++] = (unsigned)(len0 + (u - 1) * 3 * G + by_ticket[v] * lenu + j);
                }
        }
        /* B's blocks have no tickets: each is the TWIN of A's block over the same bytes, decoded by the same workgroup right behind it, from the
         * same record table */
        if (twins || tables) while (t < ne) { ord[t++] = 0xFFFFFFFFu; nt_pad++; }      /* B's blocks (tables: C's too) have no tickets in this launch */
        if (twins) {
            for (u = 1; u < nu; u++) {
                const size_t lenu = un - u * G < G ? un - u * G : G, a0 = len0 + (u - 1) * 3 * G;
                for (j = 0; j < lenu; j++) twin[a0 + j] = (unsigned)(a0 + lenu + j + 1);
            }
        }
        if (e != ne || t != ne || nt_pad != (tables ? 2 * (un - len0) : twins ? un - len0 : 0)) err = LZ4AMD_E_RUNTIME;
    }
    if (!err) {
        rc = chained_runs_create(ctx, &p->inner, (int)ne, esrc, esz, edst, ecap, efl, epre);
        if (rc) err = rc;
    }
    if (!err) {
        p->inner->dec.order = (const uint32_t*)plan_add_array(p->inner, ord, ne * sizeof *ord, &err);
        if (!err && (lz4amd_hip_h2d(LZ4AMD_CHAIN_GATE(p->inner->dec.chain, ne), gate, ne * sizeof *gate, NULL)
                     || lz4amd_hip_h2d(LZ4AMD_CHAIN_TWIN(p->inner->dec.chain, ne), twin, ne * sizeof *twin, NULL) || lz4amd_hip_sync(NULL))) err = LZ4AMD_E_RUNTIME;
    }
    if (!err && tables) {
        /* the first launch's tables: room for one per entry (only A's and unit 0's are written); the second launch: entries 2 k (B) and 2 k + 1 (C) of
         * unit k + 1, independent blocks with 64 KB in front, tables of their own that spec_gate fills from A's */
        size_t k, nB = un - len0;
        uint8_t *T, *T2 = NULL;
        hstride = lz4amd_hint_bytes((int)max_cap);
        T = (uint8_t*)plan_add_array(p->inner, NULL, hstride * ne, &err);
        if (!err) T2 = (uint8_t*)plan_add_array(p->inner, NULL, hstride * 2 * nB, &err);
        if (!err && (lz4amd_hip_memset(T, 0, hstride * ne, NULL) || lz4amd_hip_memset(T2, 0, hstride * 2 * nB, NULL))) err = LZ4AMD_E_RUNTIME;
        if (!err) q->src_size = (const int32_t*)plan_add_array(p->inner, src_sizes, un * sizeof(int), &err);
        if (!err) {
            const void** bsrc = (const void**)malloc(2 * nB * sizeof *bsrc); void** bdst = (void**)malloc(2 * nB * sizeof *bdst);
            int *bsz = (int*)malloc(2 * nB * sizeof *bsz), *bcap = (int*)malloc(2 * nB * sizeof *bcap), *bpre = (int*)malloc(2 * nB * sizeof *bpre);
            p->inner->dec.hints = T; p->inner->dec.hint_stride = hstride; p->inner->dec.hint_make = 1u;
            if (!bsrc || !bdst || !bsz || !bcap || !bpre) err = LZ4AMD_E_MEMORY;
            for (k = 0; !err && k < nB; k++) {
                const size_t e1 = len0 + 3 * k + 1;
                bsrc[2 * k] = bsrc[2 * k + 1] = esrc[e1]; bsz[2 * k] = bsz[2 * k + 1] = esz[e1]; bcap[2 * k] = bcap[2 * k + 1] = ecap[e1];
                bdst[2 * k] = edst[e1]; bdst[2 * k + 1] = edst[e1 + 1]; bpre[2 * k] = bpre[2 * k + 1] = 65536;
            }
            if (!err) {
                rc = plan_create_with_prefix(ctx, &p->inner_b, LZ4AMD_OP_DECOMPRESS, (int)(2 * nB), bsrc, bsz, bdst, bcap, bpre, 0);
                if (rc) err = rc;
                else if (lz4amd_plan_attach_hints(p->inner_b, T2, hstride)) err = LZ4AMD_E_MEMORY;
                else {
                    q->spec_result_b = (const int32_t*)p->inner_b->d_results;
                    q->tables_a = T; q->tables_b = T2; q->table_stride = hstride;
                    q->b_src_size = (int32_t*)p->inner_b->dec.src_size;            /* (written by spec_gate before the second launch reads it) */
                }
            }
            free(bsrc); free(bdst); free(bsz); free(bcap); free(bpre);
        }
    }
    free(esrc); free(edst); free(esz); free(ecap); free(epre); free(efl); free(ord); free(gate); free(twin);
    if (!err) {
        q->spec_result = (const int32_t*)p->inner->d_results;
        q->lowref = LZ4AMD_CHAIN_LOWREF(p->inner->dec.chain, ne);
        p->dec.chain = q->start;                                        /* (marks the plan as one of dependent blocks: no tables, lz4amd_plan_attach_hints) */
        p->grid = p->inner->grid;
        if (lz4amd_hip_launch_spec_fill(q, NULL) || lz4amd_hip_sync(NULL)) err = LZ4AMD_E_RUNTIME;
    }
    if (err) { lz4amd_plan_destroy(p); return err; }
    *out = p;
    return LZ4AMD_OK;
}

int lz4amd_plan_create_decompress_chained(lz4amd_ctx* ctx, lz4amd_plan** out, int n,
                                          const void* const* d_src, const int* src_sizes,
                                          void* d_dst0, const int* dst_caps, const unsigned char* stored, int initial_prefix)
{
    void** dsts; int* pre; unsigned char* fl; int rc, i;
    if (!ctx || !out || n <= 0 || !d_src || !src_sizes || !dst_caps || !d_dst0 || initial_prefix < 0) return LZ4AMD_E_ARG;
    *out = NULL;
    /* two units and more: side by side; LZ4AMD_CHAIN_SERIAL=1 keeps the chain of copy stages (also what a plan falls back to) */
    if (n >= 2 && !getenv("LZ4AMD_CHAIN_SERIAL") && chained_spec_create(ctx, out, n, d_src, src_sizes, d_dst0, dst_caps, stored, initial_prefix) == LZ4AMD_OK)
        return LZ4AMD_OK;
    dsts = (void**)malloc((size_t)n * sizeof *dsts); pre = (int*)calloc((size_t)n, sizeof *pre); fl = (unsigned char*)calloc((size_t)n, 1);
    if (!dsts || !pre || !fl) { free(dsts); free(pre); free(fl); return LZ4AMD_E_MEMORY; }
    for (i = 0; i < n; i++) { dsts[i] = d_dst0; pre[i] = initial_prefix; fl[i] = stored && stored[i] ? 1 : 0; }
    rc = chained_runs_create(ctx, out, n, d_src, src_sizes, dsts, dst_caps, stored ? fl : NULL, pre);
    free(dsts); free(pre); free(fl);
    return rc;
}

size_t lz4amd_hint_bytes(int src_size)
{   /* header + room for one row (8 bytes) per 128 bytes of source (a row is 8 sequences: blocks that average less than 16 bytes per
     * sequence get no table) + the end row (lz4amd_params.h), a multiple of 16 */
    if (src_size < 0) return 0;
    return (LZ4AMD_HINT_HEAD + LZ4AMD_HINT_ROW * (((size_t)src_size + 127u) / 128u + 2u) + 15u) & ~(size_t)15u;
}

int lz4amd_plan_attach_hints(lz4amd_plan* p, void* d_hints, size_t stride)
{
    if (!p || ((size_t)d_hints & 15u) || (stride & 15u) || (d_hints && stride < 32)) return LZ4AMD_E_ARG;
    if (p->op == LZ4AMD_OP_COMPRESS) { p->comp.hints = (uint8_t*)d_hints; p->comp.hint_stride = stride; }
    else if (p->op == LZ4AMD_OP_COMPRESS_HC) { p->hc.hints = (uint8_t*)d_hints; p->hc.hint_stride = stride; }
    else if (p->op == LZ4AMD_OP_DECOMPRESS && !p->dec.chain) {
        if (d_hints && !p->dec.hint_stats) {
            int k, err = 0;
            for (k = 0; k < LZ4AMD_PLAN_MAX_BUFS && p->bufs[k]; k++) {}
            if (k >= LZ4AMD_PLAN_MAX_BUFS) return LZ4AMD_E_MEMORY;
            (void)lz4amd_hip_use_device(p->ctx->device);
            p->bufs[k] = lz4amd_hip_malloc(64);
            if (!p->bufs[k]) return LZ4AMD_E_MEMORY;
            if (lz4amd_hip_memset(p->bufs[k], 0, 64, NULL) || lz4amd_hip_sync(NULL)) err = LZ4AMD_E_RUNTIME;
            if (err) { lz4amd_set_error(lz4amd_hip_errstr()); return err; }
            p->dec.hint_stats = (uint32_t*)p->bufs[k];
        }
        p->dec.hints = (const uint8_t*)d_hints; p->dec.hint_stride = stride;
    }
    else return LZ4AMD_E_ARG;
    return LZ4AMD_OK;
}

int lz4amd_plan_hint_stats(lz4amd_plan* p, unsigned* used, unsigned* rejected)
{   /* since the tables were attached: blocks decoded from their table / tables that did not fit (those blocks were decoded without) */
    unsigned v[2] = {0, 0};
    if (!p || p->op != LZ4AMD_OP_DECOMPRESS || !p->dec.hint_stats) return LZ4AMD_E_ARG;
    (void)lz4amd_hip_use_device(p->ctx->device);
    if (lz4amd_hip_d2h(v, p->dec.hint_stats, sizeof v, NULL) || lz4amd_hip_sync(NULL)) { lz4amd_set_error(lz4amd_hip_errstr()); return LZ4AMD_E_RUNTIME; }
    if (used) *used = v[0];
    if (rejected) *rejected = v[1];
    return LZ4AMD_OK;
}

int lz4amd_plan_chain_stats(lz4amd_plan* p, unsigned long long out[4])
{   /* linked blocks side by side, after a launch: { units, units decoded a third time, bytes of units 1.. that lie before their unit's last
     * history-dependent byte (what the patch pass walks), decoded bytes of units 1.. }; synchronises the device */
    size_t nu, k;
    int32_t *ld = NULL, *sz = NULL; uint8_t* th = NULL;
    int rc = LZ4AMD_OK;
    if (!p || !out || !p->inner) return LZ4AMD_E_ARG;
    nu = p->spec.n_units;
    ld = (int32_t*)malloc(nu * sizeof *ld); sz = (int32_t*)malloc(nu * sizeof *sz); th = (uint8_t*)malloc(nu);
    if (!ld || !sz || !th) rc = LZ4AMD_E_MEMORY;
    (void)lz4amd_hip_use_device(p->ctx->device);
    if (!rc && (lz4amd_hip_d2h(ld, p->spec.lastdep, nu * sizeof *ld, NULL) || lz4amd_hip_d2h(sz, p->spec.size, nu * sizeof *sz, NULL)
                || lz4amd_hip_d2h(th, p->spec.three, nu, NULL) || lz4amd_hip_sync(NULL))) { lz4amd_set_error(lz4amd_hip_errstr()); rc = LZ4AMD_E_RUNTIME; }
    if (!rc && getenv("LZ4AMD_CHAIN_DEBUG") && p->inner_b) {
        unsigned u2[2] = {0, 0};
        (void)lz4amd_plan_hint_stats(p->inner_b, &u2[0], &u2[1]);
        fprintf(stderr, "lz4amd: second copies decoded from the first ones' tables: %u, tables rejected: %u (all launches)\n", u2[0], u2[1]);
    }
    if (!rc && getenv("LZ4AMD_CHAIN_DEBUG")) {          /* developer aid: blocks decoded from their twin's record table */
        size_t ne = p->inner->dec.n_blocks, cnt = 0;
        uint32_t* cw = (uint32_t*)malloc(ne * 16);
        if (cw && !lz4amd_hip_d2h(cw, LZ4AMD_CHAIN_CARRY(p->inner->dec.chain, ne), ne * 16, NULL) && !lz4amd_hip_sync(NULL)) {
            for (k = 0; k < ne; k++) cnt += cw[4 * k] == LZ4AMD_CHAIN_CARRIED;
            fprintf(stderr, "lz4amd: %zu of %zu entries decoded from their twin's record table\n", cnt, ne);
        }
        free(cw);
    }
    if (!rc) {
        out[0] = nu; out[1] = out[2] = out[3] = 0;
        for (k = 1; k < nu; k++) { out[1] += th[k] ? 1u : 0u; out[2] += (unsigned long long)(ld[k] + 1); out[3] += (unsigned long long)(sz[k] > 0 ? sz[k] : 0); }
    }
    free(ld); free(sz); free(th);
    return rc;
}

int lz4amd_plan_make_hints(lz4amd_plan* p, int on)
{   /* a decompress plan with tables attached: blocks whose table is missing or unusable get theirs written while they are decoded */
    if (!p || p->op != LZ4AMD_OP_DECOMPRESS || p->dec.chain || (on && !p->dec.hints)) return LZ4AMD_E_ARG;
    p->dec.hint_make = on ? 1u : 0u;
    return LZ4AMD_OK;
}

int lz4amd_plan_hints_made(lz4amd_plan* p, unsigned* made)
{
    unsigned v[3] = {0, 0, 0};
    if (!p || p->op != LZ4AMD_OP_DECOMPRESS || !p->dec.hint_stats || !made) return LZ4AMD_E_ARG;
    (void)lz4amd_hip_use_device(p->ctx->device);
    if (lz4amd_hip_d2h(v, p->dec.hint_stats, sizeof v, NULL) || lz4amd_hip_sync(NULL)) { lz4amd_set_error(lz4amd_hip_errstr()); return LZ4AMD_E_RUNTIME; }
    *made = v[2];
    return LZ4AMD_OK;
}

int lz4amd_plan_set_acceleration(lz4amd_plan* p, int acceleration)
{
    if (!p || p->op != LZ4AMD_OP_COMPRESS) return LZ4AMD_E_ARG;
    p->comp.acceleration = acceleration < 1 ? 1 : (acceleration > 65537 ? 65537 : acceleration);     /* lz4.c:1386-1387 */
    return LZ4AMD_OK;
}

static int launch_stage(lz4amd_plan* p, int stage, void* stream)
{
    if (p->inner) return stage == 0 ? lz4amd_hip_launch_spec(&p->spec, &p->inner->dec, p->inner->grid, p->inner_b ? &p->inner_b->dec : NULL, p->inner_b ? p->inner_b->grid : 0u, p->spec_max_cap, stream) : 0;
    if (p->op == LZ4AMD_OP_DECOMPRESS)
        return stage == 0 ? lz4amd_hip_launch_decompress(&p->dec, p->grid, stream) : 0;
    if (p->op == LZ4AMD_OP_XXH32) return stage == 0 ? lz4amd_hip_launch_xxh32(&p->xxh, stream) : 0;
    if (p->op == LZ4AMD_OP_GATHER) return stage == 0 ? lz4amd_hip_launch_gather(&p->gather, stream) : 0;
    if (p->op == LZ4AMD_OP_COMPRESS_HC) return stage == 0 ? lz4amd_hip_launch_compress_hc(&p->hc, p->grid, stream) : 0;
    return stage == 0 ? lz4amd_hip_launch_compress(&p->comp, p->grid, stream) : 0;
}
static int n_stages(const lz4amd_plan* p) { (void)p; return 1; }

int lz4amd_plan_launch(lz4amd_plan* p, void* stream)
{
    int s;
    if (!p) return LZ4AMD_E_ARG;
    if (lz4amd_hip_use_device(p->ctx->device)) { lz4amd_set_error(lz4amd_hip_errstr()); return LZ4AMD_E_RUNTIME; }
    for (s = 0; s < n_stages(p); s++)
        if (launch_stage(p, s, stream)) { lz4amd_set_error(lz4amd_hip_errstr()); return LZ4AMD_E_RUNTIME; }
    return LZ4AMD_OK;
}

int lz4amd_plan_launch_timed(lz4amd_plan* p, void* stream, float kernel_ms[4], float* total_ms)
{
    int s, ns;
    if (!p) return LZ4AMD_E_ARG;
    if (lz4amd_hip_use_device(p->ctx->device)) { lz4amd_set_error(lz4amd_hip_errstr()); return LZ4AMD_E_RUNTIME; }
    ns = n_stages(p);
    for (s = 0; s <= ns; s++)
        if (!p->ev[s] && !(p->ev[s] = lz4amd_hip_event_create())) { lz4amd_set_error("hipEventCreate failed"); return LZ4AMD_E_RUNTIME; }
    for (s = 0; s < ns; s++) {
        if (lz4amd_hip_event_record(p->ev[s], stream) || launch_stage(p, s, stream)) {
            lz4amd_set_error(lz4amd_hip_errstr()); return LZ4AMD_E_RUNTIME;
        }
    }
    if (lz4amd_hip_event_record(p->ev[ns], stream) || lz4amd_hip_event_sync(p->ev[ns])) {
        lz4amd_set_error(lz4amd_hip_errstr()); return LZ4AMD_E_RUNTIME;
    }
    for (s = 0; s < 4; s++) if (kernel_ms) kernel_ms[s] = s < ns ? lz4amd_hip_event_ms(p->ev[s], p->ev[s + 1]) : 0.f;
    if (total_ms) *total_ms = lz4amd_hip_event_ms(p->ev[0], p->ev[ns]);
    return LZ4AMD_OK;
}

int lz4amd_plan_profile(lz4amd_plan* p, unsigned long long* words, int max_words)
{   /* 8 words per workgroup: start stamp, stamp after the pre-parse, cycles in EMIT/LOAD/INDEX, cycles in COPY, end stamp; nseq, total, csize */
    int n;
    const uint64_t* src;
    if (!p) return 0;
    if (p->inner) return lz4amd_plan_profile(p->inner, words, max_words);
    src = p->op == LZ4AMD_OP_DECOMPRESS ? p->dec.prof : p->op == LZ4AMD_OP_COMPRESS_HC ? p->hc.prof : p->comp.prof;
    if (!src) return 0;
    (void)lz4amd_hip_use_device(p->ctx->device);
    n = (int)p->grid * 8;
    if (p->op == LZ4AMD_OP_DECOMPRESS && max_words > n) n += LZ4AMD_TRACE_BYTES / 8;      /* (the developer build's event trace behind the workgroups' words) */
    if (n > max_words) n = max_words;
    if (lz4amd_hip_d2h(words, src, (size_t)n * 8, NULL) || lz4amd_hip_sync(NULL)) return 0;
    return n;
}

const int* lz4amd_plan_device_results(const lz4amd_plan* p) { return p ? p->d_results : NULL; }

int lz4amd_plan_results(lz4amd_plan* p, int* results, void* stream)
{
    if (!p || (p->n && !results)) return LZ4AMD_E_ARG;
    (void)lz4amd_hip_use_device(p->ctx->device);
    if (lz4amd_hip_d2h(results, p->d_results, (size_t)p->n * sizeof(int), stream) || lz4amd_hip_sync(stream)) {
        lz4amd_set_error(lz4amd_hip_errstr());
        return LZ4AMD_E_RUNTIME;
    }
    return LZ4AMD_OK;
}

static int one_shot(lz4amd_ctx* ctx, lz4amd_op op, const void* const* d_src, const int* src_sizes,
                    void* const* d_dst, const int* dst_caps, int* results, int n, void* stream)
{
    lz4amd_plan* p = NULL;
    int rc = lz4amd_plan_create(ctx, &p, op, n, d_src, src_sizes, d_dst, dst_caps, 0);
    if (rc) return rc;
    rc = lz4amd_plan_launch(p, stream);
    if (!rc) rc = lz4amd_plan_results(p, results, stream);
    lz4amd_plan_destroy(p);
    return rc;
}

int lz4amd_compress_batch(lz4amd_ctx* ctx, const void* const* d_src, const int* src_sizes,
                          void* const* d_dst, const int* dst_caps, int* results, int n, void* stream)
{ return one_shot(ctx, LZ4AMD_OP_COMPRESS, d_src, src_sizes, d_dst, dst_caps, results, n, stream); }

int lz4amd_compress_hc_batch(lz4amd_ctx* ctx, const void* const* d_src, const int* src_sizes,
                             void* const* d_dst, const int* dst_caps, int* results, int n, int level, void* stream)
{
    lz4amd_plan* p = NULL;
    int rc = lz4amd_plan_create(ctx, &p, LZ4AMD_OP_COMPRESS_HC, n, d_src, src_sizes, d_dst, dst_caps, level);
    if (rc) return rc;
    rc = lz4amd_plan_launch(p, stream);
    if (!rc) rc = lz4amd_plan_results(p, results, stream);
    lz4amd_plan_destroy(p);
    return rc;
}

int lz4amd_decompress_batch(lz4amd_ctx* ctx, const void* const* d_src, const int* src_sizes,
                            void* const* d_dst, const int* dst_caps, int* results, int n, void* stream)
{ return one_shot(ctx, LZ4AMD_OP_DECOMPRESS, d_src, src_sizes, d_dst, dst_caps, results, n, stream); }

int lz4amd_plan_set_row0(lz4amd_plan* p, int src_size, int dst_cap, int level, void* stream)
{
    const int32_t* dsz; const int32_t* dcap;
    if (!p || p->n != 1) return LZ4AMD_E_ARG;
    if (p->op == LZ4AMD_OP_DECOMPRESS) { dsz = p->dec.src_size; dcap = p->dec.dst_cap; }
    else if (p->op == LZ4AMD_OP_COMPRESS_HC) { dsz = p->hc.src_size; dcap = p->hc.dst_cap; p->hc.level = hc_level_norm(level); }
    else if (p->op == LZ4AMD_OP_COMPRESS) { dsz = p->comp.src_size; dcap = p->comp.dst_cap; }
    else return LZ4AMD_E_ARG;
    p->row0[0] = src_size; p->row0[1] = dst_cap;          /* (the copies are asynchronous: the source must outlive the call) */
    if (lz4amd_hip_h2d((void*)dsz, &p->row0[0], sizeof(int), stream) || lz4amd_hip_h2d((void*)dcap, &p->row0[1], sizeof(int), stream)) {
        lz4amd_set_error(lz4amd_hip_errstr());
        return LZ4AMD_E_RUNTIME;
    }
    return LZ4AMD_OK;
}

void lz4amd_plan_set_level(lz4amd_plan* p, int level)
{   /* LZ4_compress_HC: the compression level; LZ4_compress_fast: the acceleration */
    if (p && p->op == LZ4AMD_OP_COMPRESS_HC) { p->hc.level = hc_level_norm(level); p->level = level; }
    else if (p && p->op == LZ4AMD_OP_COMPRESS) (void)lz4amd_plan_set_acceleration(p, level);
}

int lz4amd_plan_bind_host_row(lz4amd_plan* p, int* row)
{
    if (!p || p->n != 1 || !row) return LZ4AMD_E_ARG;
    /* (a plan made with a history column reads the history's length from row[3]) */
    if (p->op == LZ4AMD_OP_DECOMPRESS) { p->dec.src_size = row; p->dec.dst_cap = row + 1; p->dec.result = row + 2; if (p->dec.prefix) p->dec.prefix = row + 3; }
    else if (p->op == LZ4AMD_OP_COMPRESS_HC) { p->hc.src_size = row; p->hc.dst_cap = row + 1; p->hc.result = row + 2; if (p->hc.prefix) p->hc.prefix = row + 3; }
    else if (p->op == LZ4AMD_OP_COMPRESS) { p->comp.src_size = row; p->comp.dst_cap = row + 1; p->comp.result = row + 2; if (p->comp.prefix) p->comp.prefix = row + 3; }
    else return LZ4AMD_E_ARG;
    p->d_results = row + 2;
    return LZ4AMD_OK;
}

/* ------------------------------------------------------------------ calibration */
int lz4amd_stream_copy_ms(lz4amd_ctx* ctx, void* d_dst, const void* d_src, size_t bytes, int reps, void* stream, float* best_ms)
{
    void *e0, *e1;
    float best = -1.f;
    int r, shape;
    if (!ctx || !d_dst || !d_src || !best_ms || bytes < 16 || reps < 1) return LZ4AMD_E_ARG;
    (void)lz4amd_hip_use_device(ctx->device);
    e0 = lz4amd_hip_event_create(); e1 = lz4amd_hip_event_create();
    if (!e0 || !e1) { lz4amd_hip_event_destroy(e0); lz4amd_hip_event_destroy(e1); return LZ4AMD_E_RUNTIME; }
    /* the best of a few shapes of the same copy (granules in flight per lane, non-temporal or not, workgroups per CU): the
       calibration is this box's stream rate, not one kernel's */
    for (shape = 0; shape < 15; shape++) {
        const unsigned variant = (unsigned)(shape % 5), per_cu = shape < 5 ? 8u : shape < 10 ? 16u : 32u;
        for (r = 0; r < reps + 1; r++) {                   /* one untimed warm-up launch per shape */
            float ms;
            if (lz4amd_hip_event_record(e0, stream) || lz4amd_hip_launch_stream_copy(d_dst, d_src, bytes, (unsigned)ctx->n_cus * per_cu, variant, stream)
                || lz4amd_hip_event_record(e1, stream) || lz4amd_hip_event_sync(e1)) {
                lz4amd_set_error(lz4amd_hip_errstr());
                lz4amd_hip_event_destroy(e0); lz4amd_hip_event_destroy(e1);
                return LZ4AMD_E_RUNTIME;
            }
            ms = lz4amd_hip_event_ms(e0, e1);
            if (r > 0 && ms > 0.f && (best < 0.f || ms < best)) best = ms;
        }
    }
    lz4amd_hip_event_destroy(e0); lz4amd_hip_event_destroy(e1);
    *best_ms = best;
    return best > 0.f ? LZ4AMD_OK : LZ4AMD_E_RUNTIME;
}
