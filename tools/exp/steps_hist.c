/* throw-away: distribution of hash-chain walk lengths per position (level 9, cap 250) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
int lz4amd_datagen(void* buf, size_t size, double match_p, double lit_p, uint32_t seed);
static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static long g_cnt_iters, g_cnt_calls;
int main(int argc, char** argv) {
    int CAP = argc > 2 ? atoi(argv[2]) : 250;
    int n = 262144, pct = argc > 1 ? atoi(argv[1]) : 60;
    uint8_t* s = malloc(n); lz4amd_datagen(s, n, pct / 100.0, 0, 0);
    static int head[32768]; int* chain = malloc(n * 4); int* steps = calloc(n, 4);
    for (int i = 0; i < 32768; i++) head[i] = -1;
    for (int p = 0; p + 4 <= n; p++) { uint32_t h = (rd32(s + p) * 2654435761u) >> 17; chain[p] = head[h]; head[h] = p; }
    long tot = 0; int hist[10] = {0};
    for (int p = 0; p <= n - 12; p++) {
        int best = 0, att = 256, q = chain[p], lim = n - 5 - p; if (lim > CAP) lim = CAP;
        while (q >= 0 && p - q <= 65535 && att-- > 0) {
            steps[p]++;
            if (s[q + best] == s[p + best] && rd32(s + q) == rd32(s + p)) { int l = 4; while (l < lim && s[q + l] == s[p + l]) l++; g_cnt_calls++; g_cnt_iters += (l + 3) / 4; if (l > best) { best = l; if (l >= lim) break; } }
            q = chain[q];
        }
        tot += steps[p];
        int b = steps[p] == 0 ? 0 : steps[p] <= 4 ? 1 : steps[p] <= 16 ? 2 : steps[p] <= 64 ? 3 : steps[p] < 256 ? 4 : 5; hist[b]++;
    }
    printf("cap %d: count calls/pos %.2f, 4-byte iterations/pos %.2f\n", CAP, (double)g_cnt_calls / n, (double)g_cnt_iters / n); printf("P%d avg %.2f  hist 0:%d <=4:%d <=16:%d <=64:%d <256:%d 256:%d\n", pct, (double)tot / n, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5]);
    /* max over lanes of per-lane sums for tiles of 8192 (8 per lane) */
    long summax = 0, sumavg = 0; int tiles = 0;
    for (int t0 = 0; t0 < n; t0 += 8192, tiles++) {
        for (int w = 0; w < 16; w++) { int mx = 0; long sm = 0;
            for (int l = 0; l < 64; l++) { int sum = 0; for (int k = 0; k < 8; k++) { int p = t0 + k * 1024 + w * 64 + l; if (p < n) sum += steps[p]; } if (sum > mx) mx = sum; sm += sum; }
            summax += mx; sumavg += sm / 64; }
    }
    printf("  per wave-tile: mean of max-lane steps %.1f, mean of avg-lane steps %.1f\n", (double)summax / (tiles * 16), (double)sumavg / (tiles * 16));
    /* deep-first heuristic: queued if first delta < 512 */
    { long qn = 0, qsteps = 0, smax = 0, sbulk = 0; int tl = 0;
      for (int t0 = 0; t0 < n; t0 += 8192, tl++) {
        long lanesum[1024]; memset(lanesum, 0, sizeof lanesum); long q = 0, qs = 0, qmax = 0;
        for (int pp = 0; pp < 8192 && t0 + pp < n; pp++) { int p = t0 + pp; int d = chain[p] >= 0 ? p - chain[p] : 0;
            if (d && d < 512) { q++; qs += steps[p]; if (steps[p] > qmax) qmax = steps[p]; } else lanesum[pp & 1023] += steps[p]; }
        long mx = 0, tot = 0; for (int l = 0; l < 1024; l++) { if (lanesum[l] > mx) mx = lanesum[l]; tot += lanesum[l]; }
        qn += q; qsteps += qs; smax += mx; sbulk += tot / 1024;
        if (tl < 3) printf("  tile %d: queued %ld (steps %ld, max %ld), static max-lane %ld avg-lane %ld\n", tl, q, qs, qmax, mx, tot / 1024);
      }
      printf("  per tile: queued %.0f positions with %.0f steps; static max-lane %.0f, avg-lane %.0f\n", (double)qn / tl, (double)qsteps / tl, (double)smax / tl, (double)sbulk / tl); }
    return 0;
}
