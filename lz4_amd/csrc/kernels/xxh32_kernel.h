// xxh32_kernel.h -- batched XXH32 (seed 0) over a table of independent blocks, for gfx950.
//
// What lz4frame uses as block checksum and content checksum (lib/xxhash.c:392 XXH32 ->
// XXH32_endian_align xxhash.c:352-389; round xxhash.c:269-275; tail + avalanche xxhash.c:291-348;
// lz4frame.c:904 block checksum, seed 0).  XXH32 is four independent 32-bit recurrences
//     v[i] = rotl(v[i] + word * P2, 13) * P1        over every fourth dword,
// each strictly serial (the rotate mixes carries, so there is no combine operator): a block cannot
// be split.  The parallelism is ACROSS blocks: one wave per block.  The wave streams the block with
// coalesced 16-byte loads (1 KB per wave-load, the next kilobyte prefetched into registers while
// the current one is consumed), parks the kilobyte in LDS, and lanes 0..3 run the four recurrences
// over its 64 stripes.  Bound: HBM in the limit (reads the block once, writes 4 bytes); today the
// dependent multiply chain (~4 instructions per 16 bytes per lane) sets the pace.
#pragma once
#include "lz4_common.h"
#include "../lz4amd_params.h"

namespace lz4amd {

using XxhBatch = ::lz4amd_xxh_params;

enum : uint32_t {
    kXxhP1 = 0x9E3779B1u, kXxhP2 = 0x85EBCA77u, kXxhP3 = 0xC2B2AE3Du, kXxhP4 = 0x27D4EB2Fu, kXxhP5 = 0x165667B1u,
    kXxhChunk = 1024,                   // bytes per wave-load (64 stripes of 16)
};
__device__ __forceinline__ uint32_t rotl32(uint32_t x, uint32_t r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint32_t xxh_round(uint32_t v, uint32_t w) { return rotl32(v + w * kXxhP2, 13) * kXxhP1; }

__device__ __forceinline__ void xxh32_block_body(const XxhBatch& P) {
    LZ4AMD_DYN_LDS(smem);                                   // kXxhChunk bytes
    uint32_t* stage = (uint32_t*)smem;
    const uint32_t lane = lane_id();
    const uint32_t b = blockIdx.x;
    const lz4amd_gsrc src = LZ4AMD_TO_GSRC(P.src[b]);
    const int32_t n_i = P.src_size[b];
    const uint32_t n = n_i > 0 ? (uint32_t)n_i : 0;
    const uint32_t nfull = n & ~15u;                        // bytes consumed by the stripe loop
    uint32_t v = lane == 0 ? kXxhP1 + kXxhP2 : lane == 1 ? kXxhP2 : lane == 2 ? 0u : 0u - kXxhP1;   // seed 0
    // -- stripes, one kilobyte at a time
    U32x4 cur; cur[0] = cur[1] = cur[2] = cur[3] = 0;
    if (16 * lane + 16 <= nfull) cur = ld_global16(src + 16 * lane);
    for (uint32_t base = 0; base < nfull; base += kXxhChunk) {
        wave_lds_fence();                                   // previous chunk's readers are done
        *(U32x4*)(stage + 4 * lane) = cur;
        const uint32_t nb = base + kXxhChunk + 16 * lane;
        if (nb + 16 <= nfull) cur = ld_global16(src + nb);  // prefetch the next kilobyte
        wave_lds_fence();
        uint32_t stripes = (nfull - base) >> 4; if (stripes > 64) stripes = 64;
        if (lane < 4) {
#pragma unroll 8
            for (uint32_t s = 0; s < stripes; s++) v = xxh_round(v, stage[4 * s + lane]);
        }
    }
    // -- merge, tail and avalanche (xxhash.c:291-348, 380-388): lane 0
    const uint32_t v1 = wave_readlane(v, 0), v2 = wave_readlane(v, 1), v3 = wave_readlane(v, 2), v4 = wave_readlane(v, 3);
    if (lane == 0) {
        uint32_t h = n >= 16 ? rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18) : kXxhP5;
        h += n;
        uint32_t p = nfull;
        for (; p + 4 <= n; p += 4) {
            const uint32_t w = (uint32_t)src[p] | ((uint32_t)src[p + 1] << 8) | ((uint32_t)src[p + 2] << 16) | ((uint32_t)src[p + 3] << 24);
            h = rotl32(h + w * kXxhP3, 17) * kXxhP4;
        }
        for (; p < n; p++) h = rotl32(h + (uint32_t)src[p] * kXxhP5, 11) * kXxhP1;
        h ^= h >> 15; h *= kXxhP2; h ^= h >> 13; h *= kXxhP3; h ^= h >> 16;
        P.result[b] = (int32_t)h;
    }
}

} // namespace lz4amd
