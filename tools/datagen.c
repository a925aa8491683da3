/*
 * tools/datagen.c -- synthetic compressible-data generator for tests and bench.py.
 *
 * Restates the byte stream the reference test tool emits for
 *     datagen -g<size> -P<pct> [-L<pct>] -s<seed>
 * (tests/datagen.c:101-151 RDG_genBlock, 164-188 RDG_genOut, 60-68 RDG_rand,
 *  71-88 RDG_fillLiteralDistrib; CLI defaults tests/datagencli.c:88-155), so that the
 * BASELINE.json inputs can be produced on the GPU box where /root/reference is absent.
 * Pinned by md5 against the real tool in tests/golden/datagen_md5.json.
 *
 * Model: a 32 KB history window followed by 128 KB generation blocks; every step is either
 * a noise run drawn from a skewed printable alphabet or a copy from <=32 KB back.
 * Build: gcc -O2 -shared -fPIC -o libdatagen.so datagen.c   (also a CLI with -DDATAGEN_MAIN)
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>

#define HIST   (32u << 10)
#define CHUNK  (128u << 10)
#define NSLOT  8192u

static uint32_t prng(uint32_t* s)
{   /* multiply / xor / rotate-13 generator */
    uint32_t x = *s * 2654435761U;
    x ^= 2246822519U;
    x = (x << 13) | (x >> 19);
    return *s = x;
}
static uint32_t draw15(uint32_t* s) { return (prng(s) >> 3) & 32767u; }
static uint32_t draw_len(uint32_t* s)
{   /* 7 times out of 8 a short run 0..15, otherwise 15..526 */
    if ((prng(s) >> 7) & 7) return prng(s) & 15;
    return (prng(s) & 511) + 15;
}

static void fill_alphabet(uint8_t* slot, double weight)
{   /* slots are handed out to '0','1',... wrapping inside '('..'}', each symbol
     * getting a share proportional to the slots still free */
    uint8_t lo = weight <= 0.0 ? 0 : '(', hi = weight <= 0.0 ? 255 : '}';
    uint8_t sym = weight <= 0.0 ? 0 : '0';
    uint32_t u = 0;
    while (u < NSLOT) {
        uint32_t share = (uint32_t)((double)(NSLOT - u) * weight) + 1;
        uint32_t stop = u + share < NSLOT ? u + share : NSLOT;
        while (u < stop) slot[u++] = sym;
        sym = (sym >= hi) ? lo : (uint8_t)(sym + 1);
    }
}

/* generate window[from..upto) given window[0..from) */
static void gen_span(uint8_t* w, size_t upto, size_t from, uint32_t pmatch15,
                     const uint8_t* slot, uint32_t* seed)
{
    size_t pos = from;
    if (pos == 0) { w[0] = slot[prng(seed) & (NSLOT - 1)]; pos = 1; }
    while (pos < upto) {
        if (draw15(seed) < pmatch15) {
            size_t len = draw_len(seed) + 4;
            size_t back = draw15(seed) + 1;
            size_t stop = pos + len < upto ? pos + len : upto;
            size_t m;
            if (back > pos) back = pos;
            m = pos - back;
            while (pos < stop) w[pos++] = w[m++];
        } else {
            size_t len = draw_len(seed);
            size_t stop = pos + len < upto ? pos + len : upto;
            while (pos < stop) w[pos++] = slot[prng(seed) & (NSLOT - 1)];
        }
    }
}

/* Fill buf[0..size) with the stream of `datagen -g<size> -P<100*match_p> -s<seed>`;
 * lit_p = 0 selects the tool's default (match_p / 4.5).  Returns 0, or -1 if
 * match_p >= 1 (the all-zero special mode is not restated). */
int lz4amd_datagen(void* buf, size_t size, double match_p, double lit_p, uint32_t seed)
{
    static const size_t WIN = HIST + CHUNK;
    uint8_t slot[NSLOT];
    uint8_t* w;
    uint8_t* out = (uint8_t*)buf;
    size_t done = 0;
    uint32_t pmatch15;
    if (match_p >= 1.0) return -1;
    w = (uint8_t*)malloc(WIN);
    if (!w) return -1;
    if (lit_p == 0.0) lit_p = match_p / 4.5;
    pmatch15 = (uint32_t)(32768 * match_p);
    fill_alphabet(slot, lit_p);
    gen_span(w, HIST, 0, pmatch15, slot, &seed);
    while (done < size) {
        size_t take = size - done < CHUNK ? size - done : CHUNK;
        gen_span(w, WIN, HIST, pmatch15, slot, &seed);
        memcpy(out + done, w, take);          /* the tool emits the window's FIRST 128 KB */
        done += take;
        memcpy(w, w + CHUNK, HIST);
    }
    free(w);
    return 0;
}

#ifdef DATAGEN_MAIN
int main(int argc, char** argv)
{
    size_t size = 64 << 10; double p = 0.5; uint32_t seed = 0; int i;
    for (i = 1; i < argc; i++) {
        if (!strncmp(argv[i], "-g", 2)) {
            char* e; size = strtoull(argv[i] + 2, &e, 10);
            if (*e == 'K') size <<= 10; else if (*e == 'M') size <<= 20; else if (*e == 'G') size <<= 30;
        } else if (!strncmp(argv[i], "-P", 2)) p = atof(argv[i] + 2) / 100.0;
        else if (!strncmp(argv[i], "-s", 2)) seed = (uint32_t)strtoul(argv[i] + 2, NULL, 10);
    }
    {   uint8_t* b = (uint8_t*)malloc(size ? size : 1);
        if (!b || lz4amd_datagen(b, size, p, 0.0, seed)) return 1;
        fwrite(b, 1, size, stdout);
        free(b);
    }
    return 0;
}
#endif
