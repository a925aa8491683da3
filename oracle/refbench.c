/*
 * refbench.c -- TEST/BENCH INFRASTRUCTURE: times a CPU LZ4 block codec on the same block table
 * as bench.py, on the host cores, the way programs/bench.c does (bench.c:466-480 compress loop,
 * 522-542 decompress loop, fastest-of-N, CLOCK_MONOTONIC as programs/timefn.c:88-95), with a
 * static block partition over T pthreads.
 *
 * Linked either against the real reference (oracle/_ref/liblz4_ref.so -> kind "reference") or,
 * with -DUSE_ORACLE, against the oracle restatement (kind "port").  Never part of the product.
 *
 *   refbench <threads> <nblocks> <block_bytes> <P> <seed> <reps> [hc_level [min_seconds [unique_blocks]]]
 * prints one JSON object.  The threads are created once; a timed pass is bracketed by two barriers, passes
 * are repeated for at least min_seconds (default 1) and the fastest of <reps> such loops counts; nblocks is
 * raised to the thread count so that no thread idles.  With hc_level > 0 (reference build only) the compressor is
 * LZ4_compress_HC(level) (lz4hc.c:1519), the shape of `lz4 -b9 -B5` (BASELINE configs[3]).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <stdint.h>

#ifdef USE_ORACLE
#include "lz4_oracle.h"
#define COMPRESS(s, d, n, c) lz4o_compress_default((const uint8_t*)(s), (uint8_t*)(d), n, c)
#define DECOMPRESS(s, d, n, c) lz4o_decompress_safe((const uint8_t*)(s), (uint8_t*)(d), n, c)
#define BOUND(n) lz4o_compress_bound(n)
#define KIND "port"
#else
int LZ4_compress_default(const char*, char*, int, int);
int LZ4_decompress_safe(const char*, char*, int, int);
int LZ4_compressBound(int);
int LZ4_compress_HC(const char*, char*, int, int, int);
#define HAVE_HC 1
#define COMPRESS(s, d, n, c) LZ4_compress_default((const char*)(s), (char*)(d), n, c)
#define DECOMPRESS(s, d, n, c) LZ4_decompress_safe((const char*)(s), (char*)(d), n, c)
#define BOUND(n) LZ4_compressBound(n)
#define KIND "reference"
#endif

int lz4amd_datagen(void* buf, size_t size, double match_p, double lit_p, uint32_t seed);

typedef struct {
    int t, T, nb, bs, bound, hc;
    char *src, *comp, *out; int* csz;
    volatile int* mode;                 /* 0 compress, 1 decompress, -1 exit: set by main before the start barrier */
    pthread_barrier_t *start, *end;
    int err;
} job_t;

static void one_pass(job_t* j, int mode)
{
    int b0 = (int)((long long)j->nb * j->t / j->T), b1 = (int)((long long)j->nb * (j->t + 1) / j->T), b;
    for (b = b0; b < b1; b++) {
        if (mode == 0) {
#ifdef HAVE_HC
            if (j->hc > 0) j->csz[b] = LZ4_compress_HC(j->src + (size_t)b * j->bs, j->comp + (size_t)b * j->bound, j->bs, j->bound, j->hc);
            else
#endif
            j->csz[b] = COMPRESS(j->src + (size_t)b * j->bs, j->comp + (size_t)b * j->bound, j->bs, j->bound);
            if (j->csz[b] <= 0) j->err = 1;
        } else {
            int r = DECOMPRESS(j->comp + (size_t)b * j->bound, j->out + (size_t)b * j->bs, j->csz[b], j->bs);
            if (r != j->bs) j->err = 1;
        }
    }
}

/* threads live for the whole run: a pass is bracketed by two barriers, nothing else is inside the timed region */
static void* worker(void* arg)
{
    job_t* j = (job_t*)arg;
    for (;;) {
        int mode;
        pthread_barrier_wait(j->start);
        mode = *j->mode;
        if (mode < 0) break;
        one_pass(j, mode);
        pthread_barrier_wait(j->end);
    }
    return NULL;
}

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* seconds per pass: passes are repeated until min_s has elapsed (bench.c:481-493), fastest of reps such loops */
static double timed(volatile int* mode_word, int mode, pthread_barrier_t* start, pthread_barrier_t* end, int reps, double min_s, int* passes_out)
{
    double best = 1e30; int r, total = 0;
    for (r = 0; r < reps; r++) {
        int passes = 0; double t0, t1;
        *mode_word = mode;
        t0 = now();
        do { pthread_barrier_wait(start); pthread_barrier_wait(end); passes++; t1 = now(); } while (t1 - t0 < min_s);
        if ((t1 - t0) / passes < best) best = (t1 - t0) / passes;
        total += passes;
    }
    *passes_out = total;
    return best;
}

int main(int argc, char** argv)
{
    int T, nb, bs, P, reps, t, b, nuniq, pc = 0, pd = 0; uint32_t seed; job_t base; double bc, bd, min_s; long long ctot = 0;
    static pthread_t th[1024]; static job_t jobs[1024];
    pthread_barrier_t start, end; volatile int mode_word = 0;
    if (argc < 7) { fprintf(stderr, "usage: refbench threads nblocks block_bytes P seed reps [hc_level [min_seconds [unique_blocks]]]\n"); return 2; }
    T = atoi(argv[1]); nb = atoi(argv[2]); bs = atoi(argv[3]); P = atoi(argv[4]); seed = (uint32_t)atoi(argv[5]); reps = atoi(argv[6]);
    if (T < 1) T = 1; if (T > 1024) T = 1024;
    if (nb < T) nb = T;                                  /* no idle threads */
    memset(&base, 0, sizeof base);
    base.nb = nb; base.bs = bs; base.bound = BOUND(bs);
    base.hc = argc > 7 ? atoi(argv[7]) : 0;
    min_s = argc > 8 ? atof(argv[8]) : 1.0;
    nuniq = argc > 9 ? atoi(argv[9]) : nb; if (nuniq < 1 || nuniq > nb) nuniq = nb;
#ifndef HAVE_HC
    if (base.hc > 0) { fprintf(stderr, "refbench: the oracle port has no HC compressor\n"); return 6; }
#endif
    base.src = (char*)malloc((size_t)nb * bs); base.comp = (char*)malloc((size_t)nb * base.bound); base.out = (char*)malloc((size_t)nb * bs);
    base.csz = (int*)calloc(nb, sizeof(int));
    if (!base.src || !base.comp || !base.out || !base.csz) return 3;
    /* the first nuniq blocks are the datagen stream; the rest of the table repeats them (same work per block) */
    if (lz4amd_datagen(base.src, (size_t)nuniq * bs, P / 100.0, 0.0, seed)) return 4;
    for (b = nuniq; b < nb; b++) memcpy(base.src + (size_t)b * bs, base.src + (size_t)(b % nuniq) * bs, bs);
    memset(base.comp, 0, (size_t)nb * base.bound); memset(base.out, 0, (size_t)nb * bs);      /* touch the pages outside the timed region */
    pthread_barrier_init(&start, NULL, T + 1); pthread_barrier_init(&end, NULL, T + 1);
    for (t = 0; t < T; t++) {
        jobs[t] = base; jobs[t].t = t; jobs[t].T = T; jobs[t].mode = &mode_word; jobs[t].start = &start; jobs[t].end = &end;
        if (pthread_create(&th[t], NULL, worker, &jobs[t])) return 7;
    }
    bc = timed(&mode_word, 0, &start, &end, reps, min_s, &pc);
    bd = timed(&mode_word, 1, &start, &end, reps, min_s, &pd);
    mode_word = -1; pthread_barrier_wait(&start);
    for (t = 0; t < T; t++) { pthread_join(th[t], NULL); if (jobs[t].err) base.err = 1; }
    if (base.err || memcmp(base.src, base.out, (size_t)nb * bs)) { fprintf(stderr, "refbench: round trip failed\n"); return 5; }
    for (b = 0; b < nb; b++) ctot += base.csz[b];
    printf("{\"kind\": \"%s\", \"hc_level\": %d, \"threads\": %d, \"blocks\": %d, \"unique_blocks\": %d, \"block_bytes\": %d, \"P\": %d, \"seed\": %u, "
           "\"src_bytes\": %lld, \"comp_bytes\": %lld, \"compress_s\": %.6f, \"decompress_s\": %.6f, \"passes\": [%d, %d], \"min_loop_s\": %.2f, "
           "\"compress_GBps\": %.4f, \"decompress_GBps\": %.4f, \"roundtrip_GBps\": %.4f}\n",
           KIND, base.hc, T, nb, nuniq, bs, P, seed, (long long)nb * bs, ctot, bc, bd, pc, pd, min_s,
           1e-9 * nb * bs / bc, 1e-9 * nb * bs / bd, 1e-9 * nb * bs / (bc + bd));
    return 0;
}
