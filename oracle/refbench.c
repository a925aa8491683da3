/*
 * refbench.c -- TEST/BENCH INFRASTRUCTURE: times a CPU LZ4 block codec on the same block table
 * as bench.py, on the host cores, the way programs/bench.c does (bench.c:466-480 compress loop,
 * 522-542 decompress loop, fastest-of-N, CLOCK_MONOTONIC as programs/timefn.c:88-95), with a
 * static block partition over T pthreads.
 *
 * Linked either against the real reference (oracle/_ref/liblz4_ref.so -> kind "reference") or,
 * with -DUSE_ORACLE, against the oracle restatement (kind "port").  Never part of the product.
 *
 *   refbench <threads> <nblocks> <block_bytes> <P> <seed> <reps> [hc_level]
 * prints one JSON object.  With hc_level > 0 (reference build only) the compressor is
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

typedef struct { int t, T, nb, bs, bound, mode, hc; char *src, *comp, *out; int* csz; int err; } job_t;

static void* worker(void* arg)
{
    job_t* j = (job_t*)arg;
    int b0 = (int)((long long)j->nb * j->t / j->T), b1 = (int)((long long)j->nb * (j->t + 1) / j->T), b;
    for (b = b0; b < b1; b++) {
        if (j->mode == 0) {
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
    return NULL;
}

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static double run(job_t* base, int T, int mode)
{
    pthread_t th[256]; job_t jobs[256]; int t; double t0, t1;
    for (t = 0; t < T; t++) { jobs[t] = *base; jobs[t].t = t; jobs[t].T = T; jobs[t].mode = mode; jobs[t].err = 0; }
    t0 = now();
    for (t = 0; t < T; t++) pthread_create(&th[t], NULL, worker, &jobs[t]);
    for (t = 0; t < T; t++) { pthread_join(th[t], NULL); if (jobs[t].err) base->err = 1; }
    t1 = now();
    return t1 - t0;
}

int main(int argc, char** argv)
{
    int T, nb, bs, P, reps, r; uint32_t seed; job_t j; double bc = 1e30, bd = 1e30; long long ctot = 0; int b;
    if (argc < 7) { fprintf(stderr, "usage: refbench threads nblocks block_bytes P seed reps\n"); return 2; }
    T = atoi(argv[1]); nb = atoi(argv[2]); bs = atoi(argv[3]); P = atoi(argv[4]); seed = (uint32_t)atoi(argv[5]); reps = atoi(argv[6]);
    if (T < 1) T = 1; if (T > 256) T = 256;
    memset(&j, 0, sizeof j);
    j.nb = nb; j.bs = bs; j.bound = BOUND(bs);
    j.hc = argc > 7 ? atoi(argv[7]) : 0;
#ifndef HAVE_HC
    if (j.hc > 0) { fprintf(stderr, "refbench: the oracle port has no HC compressor\n"); return 6; }
#endif
    j.src = (char*)malloc((size_t)nb * bs); j.comp = (char*)malloc((size_t)nb * j.bound); j.out = (char*)malloc((size_t)nb * bs);
    j.csz = (int*)calloc(nb, sizeof(int));
    if (!j.src || !j.comp || !j.out || !j.csz) return 3;
    if (lz4amd_datagen(j.src, (size_t)nb * bs, P / 100.0, 0.0, seed)) return 4;
    for (r = 0; r < reps; r++) { double t = run(&j, T, 0); if (t < bc) bc = t; }
    for (r = 0; r < reps; r++) { double t = run(&j, T, 1); if (t < bd) bd = t; }
    if (j.err || memcmp(j.src, j.out, (size_t)nb * bs)) { fprintf(stderr, "refbench: round trip failed\n"); return 5; }
    for (b = 0; b < nb; b++) ctot += j.csz[b];
    printf("{\"kind\": \"%s\", \"hc_level\": %d, \"threads\": %d, \"blocks\": %d, \"block_bytes\": %d, \"P\": %d, \"seed\": %u, "
           "\"src_bytes\": %lld, \"comp_bytes\": %lld, \"compress_s\": %.6f, \"decompress_s\": %.6f, "
           "\"compress_GBps\": %.4f, \"decompress_GBps\": %.4f, \"roundtrip_GBps\": %.4f}\n",
           KIND, j.hc, T, nb, bs, P, seed, (long long)nb * bs, ctot, bc, bd,
           1e-9 * nb * bs / bc, 1e-9 * nb * bs / bd, 1e-9 * nb * bs / (bc + bd));
    return 0;
}
