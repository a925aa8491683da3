/*
 * lz4_oracle.c -- TEST INFRASTRUCTURE ONLY (see lz4_oracle.h).
 *
 * CPU restatement of the reference algorithms on the hot path, written from the
 * format documents (doc/lz4_Block_format.md, doc/lz4_Frame_format.md) and from the
 * observable behaviour of lib/lz4.c, lib/xxhash.c and lib/lz4frame.c.  Each function
 * cites the reference lines whose RESULT it reproduces.  Little-endian 64-bit hosts.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks byte equality of
 * lz4o_compress_fast against LZ4_compress_fast of oracle/_ref/liblz4_ref.so, decoder
 * equality in both directions, XXH32 known answers and frame interop
 * with the reference LZ4F_* functions; tests/golden/ holds vectors made by the real
 * reference in the build container (tests/golden/make_golden.py).
 */
#include "lz4_oracle.h"
#include <string.h>

/* ------------------------------------------------------------------ helpers */
static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static void wr16le(uint8_t* p, unsigned v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void wr32le(uint8_t* p, uint32_t v)
{ p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t rd32le(const uint8_t* p)
{ return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

enum {
    MINMATCH = 4,        /* lz4.c:242 */
    MFLIMIT = 12,        /* lz4.c:246 */
    LASTLITERALS = 5,    /* lz4.c:245 */
    MAXDIST = 65535,     /* lz4.h:673 LZ4_DISTANCE_MAX */
    SMALL_LIMIT = 65536 + MFLIMIT - 1,   /* lz4.c:710 LZ4_64Klimit */
    SKIP_TRIGGER = 6     /* lz4.c:711 */
};

int lz4o_compress_bound(int n)
{   /* lz4.h:214-215 */
    if (n < 0 || (unsigned)n > (unsigned)LZ4O_MAX_INPUT_SIZE) return 0;
    return n + n / 255 + 16;
}

/* ------------------------------------------------------------- fast compress */

/* lz4.c:777-795: two hash flavours.  Small inputs (<64KB+11) index a 8192 x U16 table
 * with a 4-byte multiplicative hash; larger ones a 4096 x U32 table with a 5-byte hash
 * computed from an 8-byte little-endian read. */
static uint32_t hash_small(const uint8_t* p) { return (rd32(p) * 2654435761U) >> (32 - 13); }
static uint32_t hash_large(const uint8_t* p)
{ return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> (64 - 12)); }

typedef struct {
    uint32_t t32[4096];   /* the 16 KB state of lz4.h:718-727, viewed either way */
    int small;
} fast_table;

static uint32_t tab_hash(const fast_table* t, const uint8_t* p)
{ return t->small ? hash_small(p) : hash_large(p); }
static uint32_t tab_get(const fast_table* t, uint32_t h)
{ return t->small ? ((const uint16_t*)t->t32)[h] : t->t32[h]; }
static void tab_put(fast_table* t, uint32_t h, uint32_t pos)
{ if (t->small) ((uint16_t*)t->t32)[h] = (uint16_t)pos; else t->t32[h] = pos; }

/* lz4.c:680-703 LZ4_count: length of the common prefix of a and b, a bounded by lim */
static unsigned common_len(const uint8_t* a, const uint8_t* b, const uint8_t* lim)
{
    const uint8_t* a0 = a;
    while (a + 8 <= lim) {
        uint64_t d = rd64(a) ^ rd64(b);
        if (d) return (unsigned)(a - a0) + (unsigned)(__builtin_ctzll(d) >> 3);
        a += 8; b += 8;
    }
    while (a < lim && *a == *b) { a++; b++; }
    return (unsigned)(a - a0);
}

/* number of extra length bytes for a nibble-overflowing length (block format doc:31-63) */
static uint8_t* put_len_ext(uint8_t* op, unsigned rest)
{
    while (rest >= 255) { *op++ = 255; rest -= 255; }
    *op++ = (uint8_t)rest;
    return op;
}

int lz4o_compress_fast(const uint8_t* src, uint8_t* dst, int n, int cap, int accel)
{
    fast_table tab;
    int limited;
    const uint8_t *ip, *anchor, *iend, *search_end, *match_end;
    uint8_t *op, *olimit;

    if (accel < 1) accel = 1;                 /* lz4.c:1386-1387 */
    if (accel > 65537) accel = 65537;
    if ((unsigned)n > (unsigned)LZ4O_MAX_INPUT_SIZE) return 0;   /* lz4.c:1360 */
    limited = !(cap >= lz4o_compress_bound(n));                  /* lz4.c:1388 */
    if (n == 0) {                                                /* lz4.c:1361-1371 */
        if (limited && cap <= 0) return 0;
        dst[0] = 0;
        return 1;
    }
    memset(&tab, 0, sizeof tab);
    tab.small = (n < SMALL_LIMIT);                               /* lz4.c:1389 */

    ip = src; anchor = src; iend = src + n;
    search_end = iend - MFLIMIT + 1;          /* lz4.c:963: first position NOT searchable */
    match_end = iend - LASTLITERALS;          /* lz4.c:964 */
    op = dst; olimit = dst + (limited ? cap : 0);

    if (n >= MFLIMIT + 1) {                   /* lz4.c:1002 (LZ4_minLength = 13) */
        uint32_t next_h;
        tab_put(&tab, tab_hash(&tab, ip), 0); /* lz4.c:1005-1010 */
        ip++;
        next_h = tab_hash(&tab, ip);

        for (;;) {
            const uint8_t* cand;
            uint8_t* token;
            /* -- probe forward until a 4-byte-verified candidate appears
             *    (lz4.c:1042-1101); the stride grows by one every 64 misses */
            {   const uint8_t* probe = ip;
                unsigned stride = 1, tries = (unsigned)accel << SKIP_TRIGGER;
                for (;;) {
                    uint32_t h = next_h;
                    uint32_t here = (uint32_t)(probe - src);
                    uint32_t idx = tab_get(&tab, h);
                    ip = probe;
                    probe += stride;
                    stride = tries++ >> SKIP_TRIGGER;
                    if (probe > search_end) goto tail;           /* lz4.c:1055 */
                    next_h = tab_hash(&tab, probe);
                    tab_put(&tab, h, here);
                    if (!tab.small && idx + MAXDIST < here) continue;   /* lz4.c:1090-1093 */
                    cand = src + idx;
                    if (rd32(cand) == rd32(ip)) break;
                }
            }
            /* -- extend backwards over pending literals (lz4.c:1105-1109) */
            while (ip > anchor && cand > src && ip[-1] == cand[-1]) { ip--; cand--; }

            /* -- literal run (lz4.c:1112-1136) */
            {   unsigned ll = (unsigned)(ip - anchor);
                token = op++;
                if (limited && op + ll + (2 + 1 + LASTLITERALS) + ll / 255 > olimit) return 0;
                if (ll >= 15) { *token = 0xF0; op = put_len_ext(op, ll - 15); }
                else *token = (uint8_t)(ll << 4);
                memcpy(op, anchor, ll);
                op += ll;
            }
            for (;;) {
                /* -- offset + match length (lz4.c:1155-1226) */
                unsigned mc;
                wr16le(op, (unsigned)(ip - cand)); op += 2;
                mc = common_len(ip + MINMATCH, cand + MINMATCH, match_end);
                ip += mc + MINMATCH;
                if (limited && op + (1 + LASTLITERALS) + (mc + 240) / 255 > olimit) return 0;
                if (mc >= 15) { *token += 15; op = put_len_ext(op, mc - 15); }
                else *token += (uint8_t)mc;
                anchor = ip;
                if (ip >= search_end) goto tail;                 /* lz4.c:1233 */

                /* -- index ip-2, then test ip itself right away (lz4.c:1236-1295) */
                tab_put(&tab, tab_hash(&tab, ip - 2), (uint32_t)(ip - 2 - src));
                {   uint32_t h = tab_hash(&tab, ip);
                    uint32_t here = (uint32_t)(ip - src);
                    uint32_t idx = tab_get(&tab, h);
                    tab_put(&tab, h, here);
                    if ((tab.small || idx + MAXDIST >= here) && rd32(src + idx) == rd32(ip)) {
                        cand = src + idx;
                        token = op++; *token = 0;      /* zero literals, next match */
                        continue;
                    }
                }
                break;
            }
            next_h = tab_hash(&tab, ++ip);                        /* lz4.c:1298 */
        }
    }
tail:
    /* -- final literal run (lz4.c:1302-1329) */
    {   size_t run = (size_t)(iend - anchor);
        if (limited && op + run + 1 + (run + 255 - 15) / 255 > olimit) return 0;
        if (run >= 15) { *op++ = 0xF0; op = put_len_ext(op, (unsigned)run - 15); }
        else *op++ = (uint8_t)(run << 4);
        memcpy(op, anchor, run);
        op += run;
    }
    return (int)(op - dst);
}

int lz4o_compress_default(const uint8_t* src, uint8_t* dst, int n, int cap)
{ return lz4o_compress_fast(src, dst, n, cap, 1); }      /* lz4.c:1472 */

/* ---------------------------------------------------------------- decompress */

int lz4o_decompress_safe_prefix(const uint8_t* src, uint8_t* dst, int csize, int cap,
                                size_t prefix)
{
    const uint8_t *ip, *iend;
    uint8_t *op, *oend;
    if (src == NULL || cap < 0) return -1;                        /* lz4.c:2036 */
    ip = src; iend = src + csize; op = dst; oend = dst + cap;
    if (cap == 0) return (csize == 1 && src[0] == 0) ? 0 : -1;    /* lz4.c:2064-2068 */
    if (csize <= 0) return -1;                                    /* lz4.c:2069 */

#define FAIL() return (int)(-(ip - src)) - 1                      /* lz4.c:2443 */
    for (;;) {
        unsigned token = *ip++;
        size_t len = token >> 4;
        size_t off;
        const uint8_t* m;
        /* literal length (read_variable_length, lz4.c:1979-2014, limit iend-15) */
        if (len == 15) {
            unsigned b;
            if (iend - ip <= 15) FAIL();
            do {
                b = *ip++; len += b;
                if (iend - ip < 15) FAIL();
            } while (b == 255);
        }
        /* literals (lz4.c:2276-2330) */
        if ((size_t)(oend - op) < len + MFLIMIT || (size_t)(iend - ip) < len + (2 + 1 + LASTLITERALS)) {
            /* must be the last sequence: consume the input exactly, stay inside dst */
            if ((size_t)(iend - ip) != len || (size_t)(oend - op) < len) FAIL();
            memmove(op, ip, len);
            op += len;
            return (int)(op - dst);
        }
        memcpy(op, ip, len); ip += len; op += len;
        /* match (lz4.c:2333-2433) */
        off = (size_t)ip[0] | ((size_t)ip[1] << 8); ip += 2;
        len = token & 15;
        if (len == 15) {
            unsigned b;                   /* limit iend-LASTLITERALS+1, no initial check */
            do {
                b = *ip++; len += b;
                if (iend - ip < LASTLITERALS - 1) FAIL();
            } while (b == 255);
        }
        len += MINMATCH;
        if (off == 0) FAIL();             /* spec: 0 is invalid (Block_format.md:77-85) */
        if (off > (size_t)(op - dst) + prefix) FAIL();            /* lz4.c:2356 */
        if ((size_t)(oend - op) < len + LASTLITERALS) FAIL();      /* lz4.c:2423 */
        m = op - off;
        while (len--) *op++ = *m++;       /* byte-serial copy == overlap semantics */
    }
#undef FAIL
}

int lz4o_decompress_safe(const uint8_t* src, uint8_t* dst, int csize, int cap)
{ return lz4o_decompress_safe_prefix(src, dst, csize, cap, 0); }

/* --------------------------------------------------------------------- XXH32 */
#define XP1 0x9E3779B1U   /* xxhash.c:263-267 */
#define XP2 0x85EBCA77U
#define XP3 0xC2B2AE3DU
#define XP4 0x27D4EB2FU
#define XP5 0x165667B1U
static uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint32_t xround(uint32_t acc, uint32_t in) { return rotl(acc + in * XP2, 13) * XP1; }

uint32_t lz4o_xxh32(const void* data, size_t len, uint32_t seed)
{   /* xxhash.c:352-389 (stripes), 291-348 (tail + avalanche) */
    const uint8_t* p = (const uint8_t*)data;
    const uint8_t* end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        do {
            v1 = xround(v1, rd32le(p)); v2 = xround(v2, rd32le(p + 4));
            v3 = xround(v3, rd32le(p + 8)); v4 = xround(v4, rd32le(p + 12));
            p += 16;
        } while (p + 16 <= end);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    } else {
        h = seed + XP5;
    }
    h += (uint32_t)len;
    while (p + 4 <= end) { h = rotl(h + rd32le(p) * XP3, 17) * XP4; p += 4; }
    while (p < end) { h = rotl(h + (*p++) * XP5, 11) * XP1; }
    h ^= h >> 15; h *= XP2; h ^= h >> 13; h *= XP3; h ^= h >> 16;
    return h;
}

/* --------------------------------------------------------------------- frame */
static size_t block_size_of(int id)
{   /* lz4frame.c:333-345 */
    static const size_t sz[4] = { 64u << 10, 256u << 10, 1u << 20, 4u << 20 };
    return (id >= 4 && id <= 7) ? sz[id - 4] : 0;
}

size_t lz4o_frame_bound(size_t n, int blockSizeID, int blockChecksum, int contentChecksum)
{
    size_t bs = block_size_of(blockSizeID);
    size_t nb;
    if (!bs) return 0;
    nb = n / bs + 1;
    return 19 + n + nb * (4 + (blockChecksum ? 4 : 0)) + 4 + (contentChecksum ? 4 : 0);
}

size_t lz4o_frame_compress(uint8_t* dst, size_t cap, const uint8_t* src, size_t n,
                           int blockSizeID, int blockChecksum, int contentChecksum,
                           int contentSizeFlag)
{
    size_t bs = block_size_of(blockSizeID);
    uint8_t* op = dst;
    uint8_t* desc;
    size_t pos = 0;
    if (!bs || cap < lz4o_frame_bound(n, blockSizeID, blockChecksum, contentChecksum)) return 0;
    /* header (lz4frame.c:782-808): magic, FLG, BD, [content size], HC */
    wr32le(op, 0x184D2204U); op += 4;
    desc = op;
    *op++ = (uint8_t)((1 << 6) | (1 << 5) | ((blockChecksum & 1) << 4) |
                      ((contentSizeFlag & 1) << 3) | ((contentChecksum & 1) << 2));
    *op++ = (uint8_t)((blockSizeID & 7) << 4);
    if (contentSizeFlag) {
        uint64_t v = n; int i;
        for (i = 0; i < 8; i++) *op++ = (uint8_t)(v >> (8 * i));
    }
    *op = (uint8_t)(lz4o_xxh32(desc, (size_t)(op - desc), 0) >> 8); op++;
    /* blocks (lz4frame.c:883-909 LZ4F_makeBlock): compressed into capacity size-1,
     * stored raw with bit 31 set when that fails */
    while (pos < n) {
        size_t chunk = (n - pos < bs) ? n - pos : bs;
        int c = lz4o_compress_fast(src + pos, op + 4, (int)chunk, (int)chunk - 1, 1);
        uint32_t field;
        if (c <= 0 || (size_t)c >= chunk) {
            memcpy(op + 4, src + pos, chunk);
            c = (int)chunk; field = (uint32_t)chunk | 0x80000000U;
        } else field = (uint32_t)c;
        wr32le(op, field);
        op += 4 + c;
        if (blockChecksum) { wr32le(op, lz4o_xxh32(op - c, (size_t)c, 0)); op += 4; }
        pos += chunk;
    }
    wr32le(op, 0); op += 4;                                       /* lz4frame.c:1222 */
    if (contentChecksum) { wr32le(op, lz4o_xxh32(src, n, 0)); op += 4; }   /* 1225-1231 */
    return (size_t)(op - dst);
}

size_t lz4o_frame_decompress(uint8_t* dst, size_t cap, const uint8_t* src, size_t n,
                             size_t* consumed)
{
    const size_t ERR = (size_t)-1;
    const uint8_t* ip = src;
    const uint8_t* iend = src + n;
    uint8_t* op = dst;
    unsigned flg, bd;
    int indep, bchk, csz, cchk, did;
    size_t bs;
    uint64_t content = 0;
    const uint8_t* desc;
    /* header (lz4frame.c:1346-1437 LZ4F_decodeHeader) */
    if (n < 7 || rd32le(ip) != 0x184D2204U) return ERR;
    ip += 4; desc = ip;
    flg = *ip++; bd = *ip++;
    if ((flg >> 6) != 1) return ERR;               /* version */
    if (flg & 0x02) return ERR;                    /* reserved */
    if (bd & 0x8F) return ERR;                     /* reserved */
    indep = (flg >> 5) & 1; bchk = (flg >> 4) & 1; csz = (flg >> 3) & 1;
    cchk = (flg >> 2) & 1; did = flg & 1;
    bs = block_size_of((bd >> 4) & 7);
    if (!bs) return ERR;
    if ((size_t)(iend - ip) < (size_t)(csz ? 8 : 0) + (did ? 4 : 0) + 1) return ERR;
    if (csz) { int i; for (i = 0; i < 8; i++) content |= (uint64_t)ip[i] << (8 * i); ip += 8; }
    if (did) ip += 4;
    if (*ip != (uint8_t)(lz4o_xxh32(desc, (size_t)(ip - desc), 0) >> 8)) return ERR;
    ip++;
    /* blocks (lz4frame.c:1729-1950) */
    for (;;) {
        uint32_t field, bsize;
        if (iend - ip < 4) return ERR;
        field = rd32le(ip); ip += 4;
        if (field == 0) break;                                    /* end mark */
        bsize = field & 0x7FFFFFFFU;
        if (bsize > bs) return ERR;                               /* lz4frame.c:1737 */
        if ((size_t)(iend - ip) < (size_t)bsize + (bchk ? 4 : 0)) return ERR;
        if (bchk && rd32le(ip + bsize) != lz4o_xxh32(ip, bsize, 0)) return ERR;
        if (field & 0x80000000U) {
            if ((size_t)(dst + cap - op) < bsize) return ERR;
            memcpy(op, ip, bsize); op += bsize;
        } else {
            size_t room = (size_t)(dst + cap - op);
            size_t hist = indep ? 0 : (size_t)(op - dst);
            int r;
            if (room > bs) room = bs;                             /* lz4frame.c:1901 */
            if (hist > 65536) hist = 65536;
            r = lz4o_decompress_safe_prefix(ip, op, (int)bsize, (int)room, hist);
            if (r < 0) return ERR;
            op += r;
        }
        ip += bsize + (bchk ? 4 : 0);
    }
    if (cchk) {                                                   /* lz4frame.c:2016-2026 */
        if (iend - ip < 4) return ERR;
        if (rd32le(ip) != lz4o_xxh32(dst, (size_t)(op - dst), 0)) return ERR;
        ip += 4;
    }
    if (csz && content != (uint64_t)(op - dst)) return ERR;       /* lz4frame.c:1984 */
    if (consumed) *consumed = (size_t)(ip - src);
    return (size_t)(op - dst);
}
