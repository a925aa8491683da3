/* emu_kernels.cpp -- TEST INFRASTRUCTURE ONLY: runs the product's kernel bodies
 * (lz4_amd/csrc/kernels/*.h) on the CPU SIMT interpreter so their logic can be unit-tested
 * without a GPU.  Exposes a plain C ABI for ctypes. */
#include "platform_emu.h"
#ifdef LZ4AMD_EMU_STATS
extern "C" { extern unsigned long long lz4amd_emu_stats[16]; }
#endif
#include "../../lz4_amd/csrc/kernels/lz4_decompress_kernel.h"
#include "../../lz4_amd/csrc/kernels/lz4_compress_kernel.h"
#include "../../lz4_amd/csrc/kernels/lz4_hc_kernel.h"
#include "../../lz4_amd/csrc/kernels/xxh32_kernel.h"
#include "../../lz4_amd/csrc/kernels/gather_kernel.h"
#include <vector>
#ifdef LZ4AMD_EMU_STATS
extern "C" { unsigned long long lz4amd_emu_stats[16]; }
#endif

extern "C" int emu_decompress_batch_prefix(const uint8_t* const* src, const int32_t* src_size,
                                           uint8_t* const* dst, const int32_t* dst_cap,
                                           int32_t* result, uint32_t n, uint32_t grid, const int32_t* prefix);
extern "C" int emu_decompress_batch(const uint8_t* const* src, const int32_t* src_size,
                                    uint8_t* const* dst, const int32_t* dst_cap,
                                    int32_t* result, uint32_t n, uint32_t grid) {
    return emu_decompress_batch_prefix(src, src_size, dst, dst_cap, result, n, grid, nullptr);
}
extern "C" int emu_decompress_batch_prefix(const uint8_t* const* src, const int32_t* src_size,
                                           uint8_t* const* dst, const int32_t* dst_cap,
                                           int32_t* result, uint32_t n, uint32_t grid, const int32_t* prefix) {
    using namespace lz4amd;
    uint32_t max_c = 0;
    for (uint32_t i = 0; i < n; i++)
        if (src_size[i] > 0 && (uint32_t)src_size[i] > max_c) max_c = src_size[i];
    uint32_t max_cap = 0;
    for (uint32_t i = 0; i < n; i++) if (dst_cap[i] > 0 && (uint32_t)dst_cap[i] > max_cap) max_cap = dst_cap[i];
    if (grid == 0) grid = n < 8 ? (n ? n : 1) : 8;
    uint64_t stride = (dec_scratch_bytes(max_c, max_cap) + 15) & ~15ull;
    std::vector<uint8_t> scratch((size_t)(stride * grid + 64));
    uint32_t ticket = 0;
    DecBatch P = {};
    P.src = src; P.src_size = src_size; P.dst = dst; P.dst_cap = dst_cap; P.result = result;
    P.n_blocks = n; P.ticket = &ticket; P.prefix = prefix; P.chain = nullptr; P.stored = nullptr;
    P.scratch = (uint8_t*)(((uintptr_t)scratch.data() + 15) & ~(uintptr_t)15); P.scratch_stride = stride; P.prof = nullptr;
    std::vector<uint64_t> profbuf((size_t)grid * 8 + 8);
    if (getenv("EMU_PROF")) P.prof = profbuf.data();       // (the developer profile's stamps: their code paths run on the interpreter too)
    simt::launch(grid, kDecThreads, kDecLdsBytes, [&] { decompress_batch_body(P); });
    return 0;
}

// dependent blocks: packed output at dst0, initial_prefix bytes of history before it, stored[i] marks blocks copied as is
extern "C" int emu_decompress_chained(const uint8_t* const* src, const int32_t* src_size, uint8_t* dst0, const int32_t* dst_cap,
                                      int32_t* result, uint32_t n, uint32_t grid, int32_t initial_prefix, const uint8_t* stored) {
    using namespace lz4amd;
    uint32_t max_c = 0;
    for (uint32_t i = 0; i < n; i++) if (src_size[i] > 0 && (uint32_t)src_size[i] > max_c) max_c = src_size[i];
    uint32_t max_cap = 0;
    for (uint32_t i = 0; i < n; i++) if (dst_cap[i] > 0 && (uint32_t)dst_cap[i] > max_cap) max_cap = dst_cap[i];
    if (grid == 0) grid = n < 4 ? (n ? n : 1) : 4;
    uint64_t stride = (dec_scratch_bytes(max_c, max_cap) + 15) & ~15ull;
    std::vector<uint8_t> scratch((size_t)(stride * grid + 64));
    std::vector<uint8_t*> dsts(n ? n : 1, dst0);
    std::vector<long long> chain(LZ4AMD_CHAIN_BYTES(n) / 8, -1); chain[0] = 0;       // (the words, what the blocks report, the gates: lz4amd_params.h)
    memset(LZ4AMD_CHAIN_GATE(chain.data(), n), 0, (size_t)n * 8);
    std::vector<int32_t> pre(n ? n : 1, initial_prefix);
    uint32_t ticket = 0;
    DecBatch P = {};
    P.src = src; P.src_size = src_size; P.dst = dsts.data(); P.dst_cap = dst_cap; P.result = result;
    P.n_blocks = n; P.ticket = &ticket; P.prefix = pre.data(); P.chain = chain.data(); P.stored = stored;
    P.scratch = (uint8_t*)(((uintptr_t)scratch.data() + 15) & ~(uintptr_t)15); P.scratch_stride = stride; P.prof = nullptr;
    if (n) simt::launch(grid, kDecThreads, kDecLdsBytes, [&] { decompress_batch_body(P); });
    return 0;
}

extern "C" int emu_compress_batch_prefix(const uint8_t* const* src, const int32_t* src_size,
                                         uint8_t* const* dst, const int32_t* dst_cap,
                                         int32_t* result, uint32_t n, uint32_t grid, const int32_t* prefix);
extern "C" int emu_compress_batch(const uint8_t* const* src, const int32_t* src_size,
                                  uint8_t* const* dst, const int32_t* dst_cap,
                                  int32_t* result, uint32_t n, uint32_t grid) {
    return emu_compress_batch_prefix(src, src_size, dst, dst_cap, result, n, grid, nullptr);
}
extern "C" int emu_compress_batch_prefix(const uint8_t* const* src, const int32_t* src_size,
                                         uint8_t* const* dst, const int32_t* dst_cap,
                                         int32_t* result, uint32_t n, uint32_t grid, const int32_t* prefix) {
    using namespace lz4amd;
    if (grid == 0) grid = n < 8 ? (n ? n : 1) : 8;
    uint32_t ticket = 0;
    CompBatch P = {};
    P.src = src; P.src_size = src_size; P.dst = dst; P.dst_cap = dst_cap; P.result = result;
    P.n_blocks = n; P.ticket = &ticket; P.prof = nullptr; P.prefix = prefix;
    if (n) simt::launch(grid, kCmpThreads, kCmpLdsBytes, [&] { compress_batch_body(P); });
    return 0;
}

extern "C" int emu_xxh32_batch(const uint8_t* const* src, const int32_t* src_size, int32_t* result, uint32_t n) {
    using namespace lz4amd;
    XxhBatch P; P.src = src; P.src_size = src_size; P.result = result; P.n_blocks = n;
    if (n) simt::launch(n, 64, kXxhChunk, [&] { xxh32_block_body(P); });
    return 0;
}

extern "C" int emu_gather_batch(const uint8_t* const* src, const int32_t* src_size, uint8_t* const* dst, const int32_t* dst_cap,
                                int32_t* result, uint32_t n) {
    using namespace lz4amd;
    GatherBatch P; P.src = src; P.src_size = src_size; P.dst = dst; P.dst_cap = dst_cap; P.result = result; P.n_blocks = n;
    if (n) simt::launch(n * kGatherSlices, kGatherThreads, 0, [&] { gather_block_body(P); });
    return 0;
}

extern "C" int emu_compress_hc_batch_prefix(const uint8_t* const* src, const int32_t* src_size,
                                            uint8_t* const* dst, const int32_t* dst_cap,
                                            int32_t* result, uint32_t n, uint32_t grid, int level, const int32_t* prefix);
extern "C" int emu_compress_hc_batch(const uint8_t* const* src, const int32_t* src_size,
                                     uint8_t* const* dst, const int32_t* dst_cap,
                                     int32_t* result, uint32_t n, uint32_t grid, int level) {
    return emu_compress_hc_batch_prefix(src, src_size, dst, dst_cap, result, n, grid, level, nullptr);
}
extern "C" int emu_compress_hc_batch_prefix(const uint8_t* const* src, const int32_t* src_size,
                                            uint8_t* const* dst, const int32_t* dst_cap,
                                            int32_t* result, uint32_t n, uint32_t grid, int level, const int32_t* prefix) {
    using namespace lz4amd;
    if (grid == 0) grid = n < 8 ? (n ? n : 1) : 8;
    uint32_t ticket = 0, max_src = 0;
    for (uint32_t i = 0; i < n; i++) if (src_size[i] > 0 && (uint32_t)src_size[i] > max_src) max_src = src_size[i];
    max_src += 65536;
    const uint64_t stride = (hc_scratch_bytes(max_src) + 255) & ~255ull;
    std::vector<uint8_t> scratch((size_t)(stride * grid + 512));
    HcBatch P = {};
    P.src = src; P.src_size = src_size; P.dst = dst; P.dst_cap = dst_cap; P.result = result;
    P.n_blocks = n; P.ticket = &ticket; P.prof = nullptr; P.level = level; P.max_src = max_src; P.prefix = prefix;
    P.scratch = (uint8_t*)(((uintptr_t)scratch.data() + 255) & ~(uintptr_t)255); P.scratch_stride = stride;
    if (n) simt::launch(grid, kHcThreads, kHcLdsBytes, [&] { hc_batch_body(P); });
    return 0;
}

extern "C" int emu_compress_hc_batch_hints(const uint8_t* const* src, const int32_t* src_size, uint8_t* const* dst, const int32_t* dst_cap,
                                           int32_t* result, uint32_t n, uint32_t grid, int level, const int32_t* prefix, uint8_t* hints, uint64_t hstride) {
    using namespace lz4amd;
    if (grid == 0) grid = n < 8 ? (n ? n : 1) : 8;
    uint32_t ticket = 0, max_src = 0;
    for (uint32_t i = 0; i < n; i++) if (src_size[i] > 0 && (uint32_t)src_size[i] > max_src) max_src = src_size[i];
    max_src += 65536;
    const uint64_t stride = (hc_scratch_bytes(max_src) + 255) & ~255ull;
    std::vector<uint8_t> scratch((size_t)(stride * grid + 512));
    HcBatch P = {};
    P.src = src; P.src_size = src_size; P.dst = dst; P.dst_cap = dst_cap; P.result = result;
    P.n_blocks = n; P.ticket = &ticket; P.prof = nullptr; P.level = level; P.max_src = max_src; P.prefix = prefix; P.hints = hints; P.hint_stride = hstride;
    P.scratch = (uint8_t*)(((uintptr_t)scratch.data() + 255) & ~(uintptr_t)255); P.scratch_stride = stride;
    if (n) simt::launch(grid, kHcThreads, kHcLdsBytes, [&] { hc_batch_body(P); });
    return 0;
}

// ---- entry-point tables: the compressor writes one per block (hints + i * stride), the decoder parses from it
extern "C" int emu_compress_batch_hints(const uint8_t* const* src, const int32_t* src_size, uint8_t* const* dst, const int32_t* dst_cap,
                                        int32_t* result, uint32_t n, uint32_t grid, const int32_t* prefix, uint8_t* hints, uint64_t stride, int32_t accel) {
    using namespace lz4amd;
    if (grid == 0) grid = n < 8 ? (n ? n : 1) : 8;
    uint32_t ticket = 0;
    CompBatch P = {};
    P.src = src; P.src_size = src_size; P.dst = dst; P.dst_cap = dst_cap; P.result = result;
    P.n_blocks = n; P.ticket = &ticket; P.prof = nullptr; P.prefix = prefix; P.hints = hints; P.hint_stride = stride; P.acceleration = accel;
    if (n) simt::launch(grid, kCmpThreads, kCmpLdsBytes, [&] { compress_batch_body(P); });
    return 0;
}
extern "C" int emu_decompress_batch_hints_make(const uint8_t* const* src, const int32_t* src_size, uint8_t* const* dst, const int32_t* dst_cap,
                                               int32_t* result, uint32_t n, uint32_t grid, const int32_t* prefix, const uint8_t* hints, uint64_t stride, uint32_t* stats, uint32_t make);
extern "C" int emu_decompress_batch_hints(const uint8_t* const* src, const int32_t* src_size, uint8_t* const* dst, const int32_t* dst_cap,
                                          int32_t* result, uint32_t n, uint32_t grid, const int32_t* prefix, const uint8_t* hints, uint64_t stride, uint32_t* stats) {
    return emu_decompress_batch_hints_make(src, src_size, dst, dst_cap, result, n, grid, prefix, hints, stride, stats, 0);
}
extern "C" int emu_decompress_batch_hints_make(const uint8_t* const* src, const int32_t* src_size, uint8_t* const* dst, const int32_t* dst_cap,
                                               int32_t* result, uint32_t n, uint32_t grid, const int32_t* prefix, const uint8_t* hints, uint64_t stride, uint32_t* stats, uint32_t make) {
    using namespace lz4amd;
    uint32_t max_c = 0, max_cap = 0;
    for (uint32_t i = 0; i < n; i++) if (src_size[i] > 0 && (uint32_t)src_size[i] > max_c) max_c = src_size[i];
    for (uint32_t i = 0; i < n; i++) if (dst_cap[i] > 0 && (uint32_t)dst_cap[i] > max_cap) max_cap = dst_cap[i];
    if (grid == 0) grid = n < 8 ? (n ? n : 1) : 8;
    uint64_t sstride = (dec_scratch_bytes(max_c, max_cap) + 15) & ~15ull;
    std::vector<uint8_t> scratch((size_t)(sstride * grid + 64));
    uint32_t ticket = 0;
    DecBatch P = {};
    P.src = src; P.src_size = src_size; P.dst = dst; P.dst_cap = dst_cap; P.result = result;
    P.n_blocks = n; P.ticket = &ticket; P.prefix = prefix; P.hints = hints; P.hint_stride = stride; P.hint_stats = stats; P.hint_make = make;
    P.scratch = (uint8_t*)(((uintptr_t)scratch.data() + 15) & ~(uintptr_t)15); P.scratch_stride = sstride;
    if (n) simt::launch(grid, kDecThreads, kDecLdsBytes, [&] { decompress_batch_body(P); });
    return 0;
}

// one block; returns the per-position search results the parse read (best length | offset << 8) -- developer checks of the search
extern "C" int emu_hc_search_results(const uint8_t* src, int32_t n, uint8_t* dst, int32_t cap, int level, uint32_t* best_out) {
    using namespace lz4amd;
    uint32_t ticket = 0; const uint32_t max_src = (uint32_t)n + 65536;
    const uint64_t stride = (hc_scratch_bytes(max_src) + 255) & ~255ull;
    std::vector<uint8_t> scratch((size_t)(stride + 512));
    int32_t res = 0;
    const uint8_t* srcs[1] = { src }; uint8_t* dsts[1] = { dst };
    HcBatch P = {};
    P.src = srcs; P.src_size = &n; P.dst = dsts; P.dst_cap = &cap; P.result = &res;
    P.n_blocks = 1; P.ticket = &ticket; P.prof = nullptr; P.level = level; P.max_src = max_src; P.prefix = nullptr;
    P.scratch = (uint8_t*)(((uintptr_t)scratch.data() + 255) & ~(uintptr_t)255); P.scratch_stride = stride;
    simt::launch(1, kHcThreads, kHcLdsBytes, [&] { hc_batch_body(P); });
    memcpy(best_out, P.scratch + hc_chain_bytes(max_src), (size_t)n * 4);
    return res;
}
