// lz4_common.h -- device-side helpers shared by the LZ4 block codec kernels (gfx950).
// Included AFTER a platform header (platform_hip.h in the product build).
#pragma once
#include <stdint.h>
#include "../lz4amd_params.h"

namespace lz4amd {

// Block format constants (doc/lz4_Block_format.md; lz4.c:242-263)
enum : uint32_t {
    kMinMatch = 4,
    kMfLimit = 12,        // last match must start >= 12 bytes before the end of the block
    kLastLiterals = 5,    // last 5 bytes are always literals
    kMaxDistance = 65535,
};

// per-block status codes returned in result[] when negative
// (the reference returns -(consumed)-1, lz4.c:2443; callers only test < 0)
__device__ __forceinline__ int err_at(uint32_t pos) { return -(int)(pos & 0x7FFFFFFFu) - 1; }

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
// ... computed where it is asked for (see opaque_u32 in the platform header): the decoder's roles use this one
__device__ __forceinline__ uint32_t lane_here() { return opaque_u32(threadIdx.x) & 63u; }
__device__ __forceinline__ uint32_t wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ uint32_t ld_u8(const uint8_t* p) { return *p; }
__device__ __forceinline__ uint32_t ld_u16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ld_u64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

using U32x4 = ::lz4amd_u32x4;             // 16 bytes as four dwords (platform header)
template <class Ptr> __device__ __forceinline__ U32x4 ld_global16(Ptr p) { return ld_global16_raw(p); }
template <class Ptr> __device__ __forceinline__ void st_global16(Ptr p, const U32x4& v) { st_global16_raw(p, v); }

// ---- entry-point tables (lz4amd_params.h): what the three writers (lz4amd_k_compress, lz4amd_k_compress_hc, the decoder's stage A) store
__device__ __forceinline__ void hint_store_row(lz4amd_gdst table, uint32_t r, uint32_t tok, uint32_t out, uint32_t ord) {
    st_global8_raw(table + LZ4AMD_HINT_HEAD + LZ4AMD_HINT_ROW * (uint64_t)r, (uint64_t)((tok & 0xFFFFFFu) | (ord << 24)) | ((uint64_t)out << 32));
}
// the table is valid from here on: its end row, its first row, its header (written last)
__device__ __forceinline__ void hint_store_head(lz4amd_gdst table, uint32_t out_size, uint32_t csize, uint32_t nseq, uint32_t nrows) {
    hint_store_row(table, nrows, csize, out_size, nseq);
    hint_store_row(table, 0, 0, 0, 0);
    U32x4 h2; h2[0] = nrows; h2[1] = h2[2] = h2[3] = 0;
    st_global16(table + 16, h2);
    U32x4 h; h[0] = LZ4AMD_HINT_MAGIC; h[1] = out_size; h[2] = csize; h[3] = nseq;
    st_global16(table, h);
}

// ---- wave-level inclusive scans (64 lanes) ------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v) { return wave_incl_sum_u32(v); }
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v) { return wave_incl_max_u32(v); }

__device__ __forceinline__ uint64_t wave_readlane64(uint64_t v, uint32_t l) {
    return (uint64_t)wave_readlane((uint32_t)v, l) | ((uint64_t)wave_readlane((uint32_t)(v >> 32), l) << 32);
}
// ---- block-level exclusive sum of two values per thread (a: u32, b: u64) ---------------
// scratch: 3 * nwaves uint32 in LDS.  Returns exclusive prefixes; totals via ta, tb.
// Contains two __syncthreads(); every thread of the block must call it.
__device__ __forceinline__ uint64_t wave_incl_sum64(uint64_t v) {
    const uint32_t lane = lane_id();
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        uint64_t y = __shfl_up(v, d);
        if (lane >= d) v += y;
    }
    return v;
}
__device__ __forceinline__ void block_excl_sum2(uint32_t a, uint64_t b, uint32_t* scratch,
                                                uint32_t& ea, uint64_t& eb, uint32_t& ta, uint64_t& tb) {
    const uint32_t lane = lane_id(), w = wave_id(), nw = blockDim.x >> 6;
    const uint32_t ia = wave_incl_sum(a);
    const uint64_t ib = wave_incl_sum64(b);
    __syncthreads();                         // scratch may still be in use by a previous call
    if (lane == 63) { scratch[w] = ia; scratch[nw + 2 * w] = (uint32_t)ib; scratch[nw + 2 * w + 1] = (uint32_t)(ib >> 32); }
    __syncthreads();
    uint32_t ba = 0, sa = 0; uint64_t bb = 0, sb = 0;
    for (uint32_t i = 0; i < nw; i++) {
        const uint32_t xa = scratch[i];
        const uint64_t xb = (uint64_t)scratch[nw + 2 * i] | ((uint64_t)scratch[nw + 2 * i + 1] << 32);
        if (i < w) { ba += xa; bb += xb; }
        sa += xa; sb += xb;
    }
    ea = ba + ia - a; eb = bb + ib - b; ta = sa; tb = sb;
}

// block-level inclusive max-scan, one value per thread. scratch: nwaves uint32.
__device__ __forceinline__ uint32_t block_incl_max(uint32_t v, uint32_t* scratch) {
    const uint32_t lane = lane_id(), w = wave_id();
    uint32_t iv = wave_incl_max(v);
    __syncthreads();
    if (lane == 63) scratch[w] = iv;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t i = 0; i < w; i++) { uint32_t x = scratch[i]; if (x > base) base = x; }
    return iv > base ? iv : base;
}

} // namespace lz4amd
