/* tools/exp/hc_sim.c -- THROW-AWAY MEASUREMENT PROGRAM (not product, not oracle).
 * CPU simulation of the parallel HC scheme (per-position longest match from an exact hash chain,
 * then a forward parse over the best[] table) to measure its compression ratio against the
 * reference's LZ4_compress_HC level 9 before the kernel is written.
 *   gcc -O2 -o hc_sim hc_sim.c ../datagen.c -I/root/reference/lib ../../oracle/_ref/liblz4_ref.so
 *   ./hc_sim <file|P<pct>> [block=262144]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "lz4.h"
#include "lz4hc.h"

int lz4amd_datagen(void* buf, size_t size, double match_p, double lit_p, uint32_t seed);

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint32_t hash4(uint32_t v) { return (v * 2654435761u) >> 17; }

typedef struct { int W, A, cap, parse; int T; int skipX; } cfg_t;
static long g_steps, g_pos;

static uint16_t* g_len; static uint16_t* g_off; static int32_t* g_chain; static int32_t g_head[32768];

static void find_all(const uint8_t* s, int n, cfg_t c)
{
    int last = n - 12;   /* last position that may start a match */
    int mlimit = n - 5;
    for (int i = 0; i < 32768; i++) g_head[i] = -1;
    for (int p = 0; p < n; p++) { g_len[p] = 0; g_off[p] = 0; }
    for (int p = 0; p + 4 <= n; p++) { uint32_t h = hash4(rd32(s + p)); g_chain[p] = g_head[h]; g_head[h] = p; }
    for (int p = 0; p <= last; p++) {
        int best = 0, boff = 0, att = c.A;
        if (c.skipX && p > 0 && (p & 7) && g_len[p - 1] > c.skipX) {   /* inherit and skip the search (runs of 8) */
            int l = g_len[p - 1] - 1, o = g_off[p - 1]; int lim = mlimit - p; if (lim > c.cap) lim = c.cap;
            while (l < lim && s[p + l] == s[p + l - o]) l++;
            g_len[p] = l; g_off[p] = o; g_pos++; continue;
        }
        int q = g_chain[p];
        int lowq = 0;
        if (c.T) { int t0 = p - p % c.T; lowq = t0 + c.T + 256 - 65536; }
        g_pos++;
        while (q >= 0 && q >= lowq && p - q <= 65535 && p - q <= c.W && att-- > 0) {
            g_steps++;
            if (s[q + best] == s[p + best] && rd32(s + q) == rd32(s + p)) {
                int l = 4; int lim = mlimit - p; if (lim > c.cap) lim = c.cap;
                while (l < lim && s[q + l] == s[p + l]) l++;
                if (l > best) { best = l; boff = p - q; if (l >= lim) break; }
            }
            q = g_chain[q];
        }
        if (best >= 4) { g_len[p] = best > 65535 ? 65535 : best; g_off[p] = boff; }
    }
}

static int lenbytes(int v) { return v >= 15 ? 1 + (v - 15) / 255 : 0; }

/* emit sequence into out (or just count) */
typedef struct { uint8_t* out; int o; int anchor; } em_t;
static void emit(em_t* e, const uint8_t* s, int ip, int ml, int off)
{
    int ll = ip - e->anchor;
    uint8_t* p = e->out + e->o;
    int tl = ll >= 15 ? 15 : ll, tm = ml - 4 >= 15 ? 15 : ml - 4;
    *p++ = (uint8_t)(tl << 4 | tm);
    if (ll >= 15) { int r = ll - 15; while (r >= 255) { *p++ = 255; r -= 255; } *p++ = (uint8_t)r; }
    memcpy(p, s + e->anchor, ll); p += ll;
    *p++ = (uint8_t)off; *p++ = (uint8_t)(off >> 8);
    if (ml - 4 >= 15) { int r = ml - 19; while (r >= 255) { *p++ = 255; r -= 255; } *p++ = (uint8_t)r; }
    e->o = (int)(p - e->out);
    e->anchor = ip + ml;
}
static void emit_last(em_t* e, const uint8_t* s, int n)
{
    int ll = n - e->anchor; uint8_t* p = e->out + e->o;
    *p++ = (uint8_t)((ll >= 15 ? 15 : ll) << 4);
    if (ll >= 15) { int r = ll - 15; while (r >= 255) { *p++ = 255; r -= 255; } *p++ = (uint8_t)r; }
    memcpy(p, s + e->anchor, ll); p += ll; e->o = (int)(p - e->out);
}

/* full forward extension of a capped match */
static int extend(const uint8_t* s, int n, int p, int off, int l)
{
    int mlimit = n - 5;
    while (p + l < mlimit && s[p + l] == s[p + l - off]) l++;
    return l;
}

static int parse_block(const uint8_t* s, int n, uint8_t* out, cfg_t c)
{
    em_t e = { out, 0, 0 };
    int last = n - 12;
    int ip = 0;
    if (c.parse == 9) {   /* optimal: backward DP with longest-only (bytes) */
        int* cost = malloc((n + 1) * sizeof(int)); int* choice = malloc((n + 1) * sizeof(int));
        /* cost in bytes ignoring literal-run length extras */
        cost[n] = 0;
        for (int p = n - 1; p >= 0; p--) {
            cost[p] = cost[p + 1] + 1; choice[p] = 0;
            if (p <= last && g_len[p] >= 4) {
                int L = extend(s, n, p, g_off[p], g_len[p]);
                int lo = 4; if (L > 64) lo = L - 32;     /* limit scan */
                for (int l = L; l >= lo; l--) {
                    int cst = 3 + lenbytes(l - 4) + cost[p + l];
                    if (cst < cost[p]) { cost[p] = cst; choice[p] = l; }
                }
            }
        }
        while (ip < n) { if (choice[ip]) { emit(&e, s, ip, choice[ip], g_off[ip]); ip += choice[ip]; } else ip++; }
        free(cost); free(choice);
        emit_last(&e, s, n);
        return e.o;
    }
    while (ip <= last) {
        int ml = g_len[ip];
        if (ml < 4) { ip++; continue; }
        int off = g_off[ip];
        if (c.parse == 0) { ml = extend(s, n, ip, off, ml); emit(&e, s, ip, ml, off); ip += ml; continue; }
        if (c.parse == 5) {   /* lazy2 inside 16 strips; matches end at the strip's end */
            int strip = (n + 15) / 16; if (strip < 1024) strip = 1024;
            int se = (ip / strip + 1) * strip; if (se > n - 5) se = n - 5;
            if (ip + 4 > se) { ip++; continue; }
            if (ip + 1 <= last && g_len[ip + 1] > ml) { ip++; continue; }
            if (ip + 2 <= last && g_len[ip + 2] > ml + 1) { ip++; continue; }
            if (ml >= c.cap) ml = extend(s, n, ip, off, ml);
            if (ip + ml > se) ml = se - ip;
            emit(&e, s, ip, ml, off); ip += ml; continue;
        }
        if (c.parse == 6 || c.parse == 7) {   /* lazy2 + flexible parsing inside 16 strips */
            int strip = (n + 15) / 16; if (strip < 1024) strip = 1024;
            int se = (ip / strip + 1) * strip; if (se > n - 5) se = n - 5;
            if (ip + 4 > se) { ip++; continue; }
            if (ip + 1 <= last && g_len[ip + 1] > ml) { ip++; continue; }
            if (ip + 2 <= last && g_len[ip + 2] > ml + 1) { ip++; continue; }
            if (ml >= c.cap) ml = extend(s, n, ip, off, ml);
            if (ip + ml > se) ml = se - ip;
            /* choose the cut l in [max(4, ml-K), ml] whose successor match reaches furthest */
            { int K = c.parse == 6 ? 62 : 15; int lo = ml - K; if (lo < 4) lo = 4; int bl = ml, breach = -1;
              for (int l = ml; l >= lo; l--) { int q = ip + l; int r = q; if (q <= last && g_len[q] >= 4 && q + 4 <= se) r = q + g_len[q]; if (r > breach) { breach = r; bl = l; } }
              ml = bl; }
            emit(&e, s, ip, ml, off); ip += ml; continue;
        }
        if (c.parse == 1 || c.parse == 2) {
            if (ip + 1 <= last && g_len[ip + 1] > ml) { ip++; continue; }
            if (c.parse == 2 && ip + 2 <= last && g_len[ip + 2] > ml + 1) { ip++; continue; }
            ml = extend(s, n, ip, off, ml); emit(&e, s, ip, ml, off); ip += ml; continue;
        }
        if (c.parse == 3) {
            /* "furthest end" rule: among q in (ip, ip+ml), find the one whose match reaches furthest beyond ip+ml */
            ml = extend(s, n, ip, off, ml);
            for (;;) {
                int end1 = ip + ml, bq = -1, bend = end1 + 0;
                int qhi = end1 - 1; if (qhi > ip + 63) qhi = ip + 63; if (qhi > last) qhi = last;   /* q <= end1-1: overlapping / adjacent */
                for (int q = ip + 1; q <= qhi; q++) {
                    int l2 = g_len[q]; if (l2 < 4) continue;
                    int e2 = q + l2;
                    if (l2 > ml && e2 > bend) { bend = e2; bq = q; }     /* longer than the first and reaching further */
                }
                if (bq < 0) { emit(&e, s, ip, ml, off); ip += ml; break; }
                if (bq - ip < 4) {     /* the first match cannot be kept: drop it, restart from bq */
                    ip = bq; ml = extend(s, n, ip, g_off[ip], g_len[ip]); off = g_off[ip]; continue;
                }
                /* keep a shortened first match, continue with the second */
                emit(&e, s, ip, bq - ip, off);
                ip = bq; off = g_off[ip]; ml = extend(s, n, ip, off, g_len[ip]);
            }
            continue;
        }
        if (c.parse == 4) {
            /* as 3, but second candidate needs only to reach further by > k and the shortened first >= 4; prefer max(end) */
            ml = extend(s, n, ip, off, ml);
            for (;;) {
                int end1 = ip + ml, bq = -1, bend = end1 + 1;
                int qhi = end1 - 1; if (qhi > ip + 63) qhi = ip + 63; if (qhi > last) qhi = last;
                for (int q = ip + 1; q <= qhi; q++) {
                    int l2 = g_len[q]; if (l2 < 4) continue;
                    int e2 = q + l2;
                    if (q - ip < 4 && l2 <= ml) continue;
                    if (e2 > bend) { bend = e2; bq = q; }
                }
                if (bq < 0) { emit(&e, s, ip, ml, off); ip += ml; break; }
                if (bq - ip < 4) { ip = bq; ml = extend(s, n, ip, g_off[ip], g_len[ip]); off = g_off[ip]; continue; }
                emit(&e, s, ip, bq - ip, off);
                ip = bq; off = g_off[ip]; ml = extend(s, n, ip, off, g_len[ip]);
            }
            continue;
        }
    }
    emit_last(&e, s, n);
    return e.o;
}

int main(int argc, char** argv)
{
    size_t total = 8u << 20; int blk = argc > 2 ? atoi(argv[2]) : 262144;
    uint8_t* data;
    if (argc < 2) return 1;
    if (argv[1][0] == 'P' && argv[1][1] >= '0' && argv[1][1] <= '9') {
        data = malloc(total); lz4amd_datagen(data, total, atoi(argv[1] + 1) / 100.0, 0.0, 0);
    } else {
        FILE* f = fopen(argv[1], "rb"); if (!f) return 1;
        fseek(f, 0, SEEK_END); total = ftell(f); fseek(f, 0, SEEK_SET);
        if (total > (16u << 20)) total = 16u << 20;
        data = malloc(total); if (fread(data, 1, total, f) != total) return 1; fclose(f);
    }
    g_len = malloc(blk * 2); g_off = malloc(blk * 2); g_chain = malloc(blk * 4);
    uint8_t* out = malloc(LZ4_compressBound(blk)); uint8_t* chk = malloc(blk);
    long ref9 = 0, ref1 = 0, ref12 = 0;
    for (size_t o = 0; o < total; o += blk) {
        int n = total - o < (size_t)blk ? (int)(total - o) : blk;
        ref9 += LZ4_compress_HC((char*)data + o, (char*)out, n, LZ4_compressBound(blk), 9);
        ref12 += LZ4_compress_HC((char*)data + o, (char*)out, n, LZ4_compressBound(blk), 12);
        ref1 += LZ4_compress_default((char*)data + o, (char*)out, n, LZ4_compressBound(blk));
    }
    printf("%s: %zu bytes, blocks of %d: ref fast %ld (%.3f)  HC9 %ld (%.3f)  HC12 %ld (%.3f)\n", argv[1], total, blk, ref1, (double)total / ref1, ref9, (double)total / ref9, ref12, (double)total / ref12);
    cfg_t cfgs[] = {
        {65535, 256, 250, 5, 0, 0}, {65535, 256, 250, 5, 0, 16}, {65535, 256, 250, 5, 0, 32}, {65535, 256, 250, 5, 0, 64}, {65535, 256, 250, 5, 0, 128},
        {65535, 256, 1 << 30, 2, 0}, {65535, 256, 1<<30, 5, 0}, {65535, 256, 64, 5, 0}, {65535, 256, 64, 5, 4096}, {65535, 256, 64, 5, 2048}, {65535, 256, 32, 5, 2048}, {65535, 256, 120, 5, 2048},
        {65535, 128, 64, 5, 2048}, {65535, 64, 64, 5, 2048},
    };
    for (unsigned k = 0; k < sizeof cfgs / sizeof cfgs[0]; k++) {
        long sum = 0; g_steps = g_pos = 0;
        for (size_t o = 0; o < total; o += blk) {
            int n = total - o < (size_t)blk ? (int)(total - o) : blk;
            if (n < 13) { sum += n + 1; continue; }
            find_all(data + o, n, cfgs[k]);
            int cs = parse_block(data + o, n, out, cfgs[k]);
            int r = LZ4_decompress_safe((char*)out, (char*)chk, cs, n);
            if (r != n || memcmp(chk, data + o, n)) { printf("  cfg %u: ROUND TRIP FAILED at block %zu (r=%d)\n", k, o / blk, r); return 2; }
            sum += cs;
        }
        printf("  skip>%3d steps/pos %5.1f", cfgs[k].skipX, (double)g_steps / g_pos);
        printf("  W=%5d A=%3d cap=%4d parse=%d : %ld  ratio %.3f  vs HC9 %+.2f%%\n", cfgs[k].W, cfgs[k].A, cfgs[k].cap > 65535 ? 0 : cfgs[k].cap, cfgs[k].parse,
               sum, (double)total / sum, 100.0 * ((double)sum / ref9 - 1));
    }
    return 0;
}
