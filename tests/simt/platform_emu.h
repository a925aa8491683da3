// platform_emu.h -- TEST INFRASTRUCTURE ONLY: CPU-interpreter twin of
// lz4_amd/csrc/kernels/platform_hip.h (see simt_emu.h).
#pragma once
#include "simt_emu.h"
#include <stdint.h>
#define LZ4AMD_LDS_PTR(T) T*
#define LZ4AMD_TO_LDS_PTR(T, p) ((T*)(p))
typedef const uint8_t* lz4amd_gsrc;
typedef uint8_t* lz4amd_gdst;
#define LZ4AMD_TO_GSRC(p) ((lz4amd_gsrc)(p))
#define LZ4AMD_TO_GDST(p) ((lz4amd_gdst)(p))
static inline void lds_or_release(uint32_t* w, uint32_t bits) { *w |= bits; }
static inline uint32_t lds_load_acquire(const uint32_t* w) { return *(volatile const uint32_t*)w; }
static inline uint64_t lds_load_acquire64(const uint64_t* w) { return *(volatile const uint64_t*)w; }
static inline void lds_store_release(uint32_t* w, uint32_t v) { *(volatile uint32_t*)w = v; }
static inline void lds_store_release64(uint64_t* w, uint64_t v) { *(volatile uint64_t*)w = v; }
template <class E> static inline E lds_load_ent(const E* p) { E e; memcpy(&e, (const void*)p, sizeof(E)); return e; }
static inline void lds_load_tag_mask(const uint32_t* tagp, const uint64_t* maskp, uint32_t& tag, uint64_t& mask) {
    tag = *(const volatile uint32_t*)tagp; mask = *(const volatile uint64_t*)maskp;
}
static inline void wave_lds_fence() { (void)__ballot(1); }
static inline void wave_lds_order() { (void)__ballot(1); }
static inline void lds_store_relaxed(uint32_t* w, uint32_t v) { *(volatile uint32_t*)w = v; }
static inline uint32_t lanes_below(unsigned long long m) { return (uint32_t)__builtin_popcountll(m & ((1ull << (simt::cur()->tid & 63u)) - 1)); }
static inline uint32_t opaque_u32(uint32_t x) { return x; }
static inline void lds_dma16(const void* gsrc, void* lds_dst) { memcpy((char*)lds_dst + 16 * (threadIdx.x & 63u), gsrc, 16); }
template <int N> static inline void vmem_wait() {}
static inline void vmem_wait_all() {}
static inline void wave_priority(uint32_t) {}
static inline void spin_pause() { simt::yield_to_sched(); }
static inline long long chain_load_acquire(const long long* w) { return __atomic_load_n(w, __ATOMIC_ACQUIRE); }
static inline void chain_store_release(long long* w, long long v) { __atomic_store_n(w, v, __ATOMIC_RELEASE); }
static inline void spin_pause_long() { simt::yield_to_sched(); }
template <int N> static inline void spin_pause_n() { simt::yield_to_sched(); }
static inline void chain_wait_pause() { simt::external_wait(); }
static inline void wave_priority_high() {}
static inline uint32_t align_bytes(uint32_t hi, uint32_t lo, uint32_t sh) {
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * (sh & 3)));
}
typedef uint32_t lz4amd_u32x4 __attribute__((vector_size(16)));
static inline void lds_load_flags2(const uint8_t* p0, const uint8_t* p1, uint32_t& v0, uint32_t& v1) { v0 = *(const volatile uint8_t*)p0; v1 = *(const volatile uint8_t*)p1; }
static inline uint32_t lds_load_byte(const uint8_t* p) { return *(const volatile uint8_t*)p; }
static inline void lds_store_byte(uint8_t* p, uint32_t v) { *(volatile uint8_t*)p = (uint8_t)v; }
static inline void wake_workgroup() {}
static inline void sleep_until_woken() { simt::yield_to_sched(); }
static inline void lds_load_2v(const uint32_t* p0, const uint32_t* p1, uint32_t& v0, uint32_t& v1) { v0 = *(const volatile uint32_t*)p0; v1 = *(const volatile uint32_t*)p1; }
static inline void lds_load_words_then4(const uint32_t* w0, const uint32_t* w1, const uint8_t* const p[4], uint32_t& wv0, uint32_t& wv1, uint32_t f[4]) {
    wv0 = *(const volatile uint32_t*)w0; wv1 = *(const volatile uint32_t*)w1;
    for (int k = 0; k < 4; k++) f[k] = *(const volatile uint8_t*)p[k];
}
static inline void lds_store_flag(uint8_t* p, uint32_t v) { *(volatile uint8_t*)p = (uint8_t)v; }
static inline void lds_load_pair16(const lz4amd_u32x4* p, lz4amd_u32x4& a, lz4amd_u32x4& b) { memcpy(&a, (const void*)p, 16); memcpy(&b, (const void*)(p + 1), 16); }
static inline void lds_load_quad16(const lz4amd_u32x4* p, lz4amd_u32x4& a, lz4amd_u32x4& b, lz4amd_u32x4& c, lz4amd_u32x4& d) {
    memcpy(&a, (const void*)p, 16); memcpy(&b, (const void*)(p + 1), 16); memcpy(&c, (const void*)(p + 2), 16); memcpy(&d, (const void*)(p + 3), 16); }
static inline void lds_load_2(const uint32_t* p0, const uint32_t* p1, uint32_t& v0, uint32_t& v1) { v0 = *(const volatile uint32_t*)p0; v1 = *(const volatile uint32_t*)p1; }
static inline void lds_load_pair16_then2(const lz4amd_u32x4* p, lz4amd_u32x4& a, lz4amd_u32x4& b, const uint32_t* p0, const uint32_t* p1, uint32_t& v0, uint32_t& v1) {
    lds_load_pair16(p, a, b);
    v0 = *(const volatile uint32_t*)p0; v1 = *(const volatile uint32_t*)p1;
}
static inline lz4amd_u32x4 ld_global16_raw(const uint8_t* p) { lz4amd_u32x4 v; memcpy(&v, p, 16); return v; }
static inline void st_global16_raw(uint8_t* p, const lz4amd_u32x4& v) { memcpy(p, &v, 16); }
static inline uint64_t ld_u64_g(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline void st_global8_raw(uint8_t* p, uint64_t v) { memcpy(p, &v, 8); }
static inline uint64_t clock_ticks() { return 0; }
static inline uint32_t flag_load_agent(const uint32_t* w) { return __atomic_load_n(w, __ATOMIC_RELAXED); }
static inline void flag_store_agent(uint32_t* w, uint32_t v) { __atomic_store_n(w, v, __ATOMIC_RELAXED); }
static inline uint32_t take_ticket(uint32_t* counter) { return __atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED); }
static inline uint32_t wave_readlane(uint32_t v, uint32_t l) { return (uint32_t)__shfl((int)v, (int)l); }
static inline uint32_t wave_incl_sum_u32(uint32_t v) {
    const uint32_t lane = simt::cur()->tid & 63u;
    for (uint32_t d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(v, d); if (lane >= d) v += y; }
    return v;
}
static inline uint32_t wave_incl_max_u32(uint32_t v) {
    const uint32_t lane = simt::cur()->tid & 63u;
    for (uint32_t d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(v, d); if (lane >= d && y > v) v = y; }
    return v;
}
static inline uint32_t wave_prev_u32(uint32_t v) { const uint32_t lane = simt::cur()->tid & 63u; const uint32_t y = __shfl_up(v, 1u); return lane ? y : 0u; }
static inline uint32_t wave_next_u32(uint32_t v) { const uint32_t lane = simt::cur()->tid & 63u; const uint32_t y = __shfl_down(v, 1u); return lane < 63u ? y : 0u; }
static inline uint32_t row16_min_u32(uint32_t v) {
    for (int d = 8; d >= 1; d >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)v, d); if (y < v) v = y; }
    return v;
}
static inline void wave_converge() { (void)__ballot(1); }
static inline void lds_barrier() { __syncthreads(); }
static inline void wave_lds_fence_local() { (void)__ballot(1); }
static inline void lds_store_release_local(uint32_t* w, uint32_t v) { *(volatile uint32_t*)w = v; }
static inline void lds_or_release_local(uint32_t* w, uint32_t bits) { *w |= bits; }
static inline uint32_t lds_load_acquire_local(const uint32_t* w) { return *(volatile const uint32_t*)w; }
static inline void settle_load16(lz4amd_u32x4&) {}
static inline void touch_load16(const lz4amd_u32x4&) {}
