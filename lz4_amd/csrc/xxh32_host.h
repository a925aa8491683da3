/* Host-side XXH32 (seed 0), one-shot and streaming: frame header checksum, content checksum.
 * Restated from the published algorithm (reference copy: xxhash.c:291-389 one-shot, 437-560 streaming).
 * Little-endian byte access helpers ride along: both frame translation units use them. */
#ifndef LZ4AMD_XXH32_HOST_H
#define LZ4AMD_XXH32_HOST_H
#include <stdint.h>
#include <string.h>
#define P1 0x9E3779B1u
#define P2 0x85EBCA77u
#define P3 0xC2B2AE3Du
#define P4 0x27D4EB2Fu
#define P5 0x165667B1u
typedef struct { uint32_t v[4]; uint8_t mem[16]; uint32_t memsize; uint64_t total; } xxh32_state;
static uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static __attribute__((unused)) void wr32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static void xxh32_reset_seed(xxh32_state* s, uint32_t seed) { memset(s, 0, sizeof *s); s->v[0] = seed + P1 + P2; s->v[1] = seed + P2; s->v[2] = seed; s->v[3] = seed - P1; }
static void xxh32_reset(xxh32_state* s) { xxh32_reset_seed(s, 0); }
static void xxh32_stripe(xxh32_state* s, const uint8_t* p)
{
    int i;
    for (i = 0; i < 4; i++) s->v[i] = rotl(s->v[i] + rd32(p + 4 * i) * P2, 13) * P1;
}
static void xxh32_update(xxh32_state* s, const uint8_t* p, size_t n)
{
    s->total += n;
    if (s->memsize) {
        const size_t take = 16 - s->memsize < n ? 16 - s->memsize : n;
        memcpy(s->mem + s->memsize, p, take); s->memsize += (uint32_t)take; p += take; n -= take;
        if (s->memsize < 16) return;
        xxh32_stripe(s, s->mem); s->memsize = 0;
    }
    while (n >= 16) { xxh32_stripe(s, p); p += 16; n -= 16; }
    if (n) { memcpy(s->mem, p, n); s->memsize = (uint32_t)n; }
}
static uint32_t xxh32_digest(const xxh32_state* s)
{
    const uint8_t* p = s->mem; const uint8_t* const end = s->mem + s->memsize;
    uint32_t h = s->total >= 16 ? rotl(s->v[0], 1) + rotl(s->v[1], 7) + rotl(s->v[2], 12) + rotl(s->v[3], 18) : s->v[2] + P5;
    h += (uint32_t)s->total;
    while (p + 4 <= end) { h = rotl(h + rd32(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl(h + (*p++) * P5, 11) * P1; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}
static __attribute__((unused)) uint32_t xxh32_once(const uint8_t* p, size_t n) { xxh32_state s; xxh32_reset(&s); xxh32_update(&s, p, n); return xxh32_digest(&s); }
#endif
