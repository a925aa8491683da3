// gather_kernel.h -- batched device copy over a table of rows, for gfx950: row i copies src_size[i] bytes from
// src[i] to dst[i] (any alignment on both sides).  This is what the frame writer does between compression and
// the download: the reference appends each block behind the previous one (lz4frame.c:883-914 LZ4F_makeBlock writes
// at dstPtr and advances it); here the blocks come out of the compressor in bound-sized slots, and the sizes that
// pack them are only known afterwards - one launch moves all of them into the frame's layout, so that the frame
// leaves the device in ONE transfer instead of one per block.
// HBM bound (reads and writes every byte once); kGatherSlices workgroups share a row so that few large rows still fill the chip.
#pragma once
#include "lz4_common.h"
#include "../lz4amd_params.h"

namespace lz4amd {

using GatherBatch = ::lz4amd_gather_params;
constexpr uint32_t kGatherThreads = 256, kGatherSlices = 8, kGatherSliceMin = 16384;

__device__ __forceinline__ void gather_block_body(const GatherBatch& P) {
    const uint32_t b = blockIdx.x / kGatherSlices, slice = blockIdx.x % kGatherSlices, tid = threadIdx.x;
    const int32_t n_i = P.src_size[b];
    const uint32_t n = n_i > 0 ? (uint32_t)n_i : 0;
    if (slice == 0 && tid == 0) P.result[b] = (n_i < 0 || n_i > P.dst_cap[b]) ? -1 : n_i;
    if (n_i < 0 || n_i > P.dst_cap[b]) return;
    const lz4amd_gsrc src = LZ4AMD_TO_GSRC(P.src[b]);
    const lz4amd_gdst dst = LZ4AMD_TO_GDST(P.dst[b]);
    // this slice: [lo, hi), cut at multiples of 16 of the DESTINATION address so that the body stores are aligned
    const uint32_t head = (uint32_t)((16u - ((uintptr_t)P.dst[b] & 15u)) & 15u);
    uint32_t per = (n + kGatherSlices - 1) / kGatherSlices; if (per < kGatherSliceMin) per = kGatherSliceMin;
    per = (per + 15u) & ~15u;
    uint32_t lo = slice * per, hi = lo + per;
    if (lo) lo += head;                                   // slice 0 also owns the unaligned head
    hi += head;
    if (hi > n) hi = n;
    if (lo >= hi) return;
    uint32_t p = lo;
    if (slice == 0) {                                // head bytes up to the first aligned destination address
        const uint32_t h = head < hi ? head : hi;
        if (tid < h) dst[tid] = src[tid];
        p = h;
    }
    const uint32_t body_end = p + ((hi - p) & ~15u);
    for (uint32_t q = p + 16 * tid; q < body_end; q += 16 * kGatherThreads)
        st_global16(dst + q, ld_global16(src + q));
    for (uint32_t q = body_end + tid; q < hi; q += kGatherThreads) dst[q] = src[q];
}

} // namespace lz4amd
