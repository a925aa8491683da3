// platform_hip.h -- the gfx950 build's view of the few primitives the kernel bodies use
// beyond plain HIP builtins.  (tests/simt/platform_emu.h is the CPU-interpreter twin used
// only by the unit tests; the shipped library is built from this file alone.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t lz4amd_u32x4 __attribute__((ext_vector_type(4)));   // a first-class register quad

// All LDS lives in the dynamic region, 16-byte aligned (CDNA guide G17).
#define LZ4AMD_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]

// pointer to LDS that keeps its address space through structs and selects (a generic pointer would
// turn every access into a flat_load)
#define LZ4AMD_LDS_PTR(T) __attribute__((address_space(3))) T*
#define LZ4AMD_TO_LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
// block pointers are loaded from the block table, so the compiler cannot see that they are global:
// say so, or every access through them becomes a flat_load / flat_store
typedef __attribute__((address_space(1))) const uint8_t* lz4amd_gsrc;
typedef __attribute__((address_space(1))) uint8_t* lz4amd_gdst;
#define LZ4AMD_TO_GSRC(p) ((lz4amd_gsrc)(p))
#define LZ4AMD_TO_GDST(p) ((lz4amd_gdst)(p))

// workgroup-scope release/acquire on LDS words (waves of one workgroup hand data to each
// other through LDS without a barrier; LDS is coherent inside a workgroup).
__device__ __forceinline__ void lds_or_release(uint32_t* w, uint32_t bits) {
    __hip_atomic_fetch_or(w, bits, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t lds_load_acquire(const uint32_t* w) {
    return __hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint64_t lds_load_acquire64(const uint64_t* w) {
    return __hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store_release(uint32_t* w, uint32_t v) {
    __hip_atomic_store(w, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store_release64(uint64_t* w, uint64_t v) {
    __hip_atomic_store(w, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// one 16-byte LDS read of a {u64, u32, u32} entry (a lane's 16 bytes are read in one LDS pass)
template <class E> __device__ __forceinline__ E lds_load_ent(const E* p) {
    const lz4amd_u32x4 v = *(const volatile lz4amd_u32x4*)p; __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    E e; __builtin_memcpy(&e, &v, sizeof(E)); return e;
}
// {tag, mask} of a done entry, the tag FIRST: two LDS reads issued back to back are serviced in that order,
// so a tag that matches proves the mask was already reset for that tag's region (the writer resets the
// mask before it stores the tag)
__device__ __forceinline__ void lds_load_tag_mask(const uint32_t* tagp, const uint64_t* maskp, uint32_t& tag, uint64_t& mask) {
    tag = *(const volatile __attribute__((address_space(3))) uint32_t*)tagp;
    mask = *(const volatile __attribute__((address_space(3))) uint64_t*)maskp;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// two flag bytes another wave may be writing: read now (not hoisted, not cached), both in one trip to the LDS
__device__ __forceinline__ void lds_load_flags2(const uint8_t* p0, const uint8_t* p1, uint32_t& v0, uint32_t& v1) {
    v0 = *(const volatile __attribute__((address_space(3))) uint8_t*)p0;
    v1 = *(const volatile __attribute__((address_space(3))) uint8_t*)p1;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// one byte of LDS another lane's store may just have written / may read next (the serial copy of a short-period match)
__device__ __forceinline__ uint32_t lds_load_byte(const uint8_t* p) { return *(const volatile __attribute__((address_space(3))) uint8_t*)p; }
__device__ __forceinline__ void lds_store_byte(uint8_t* p, uint32_t v) { *(volatile __attribute__((address_space(3))) uint8_t*)p = (uint8_t)v; }
// two control words, THEN four flag bytes, read now, all in one trip to the LDS (the DS unit serves a wave's reads in issue order: what the
// bytes hold was published before whatever moves the word behind the value read here)
// two words (a lane's own two: the bells it watches), read now, one trip
__device__ __forceinline__ void lds_load_2v(const uint32_t* p0, const uint32_t* p1, uint32_t& v0, uint32_t& v1) {
    v0 = *(const volatile __attribute__((address_space(3))) uint32_t*)p0;
    v1 = *(const volatile __attribute__((address_space(3))) uint32_t*)p1;
}
__device__ __forceinline__ void lds_load_words_then4(const uint32_t* w0, const uint32_t* w1, const uint8_t* const p[4], uint32_t& wv0, uint32_t& wv1, uint32_t f[4]) {
    wv0 = *(const volatile __attribute__((address_space(3))) uint32_t*)w0;
    wv1 = *(const volatile __attribute__((address_space(3))) uint32_t*)w1;
#pragma unroll
    for (int k = 0; k < 4; k++) f[k] = *(const volatile __attribute__((address_space(3))) uint8_t*)p[k];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void lds_store_flag(uint8_t* p, uint32_t v) { *(volatile __attribute__((address_space(3))) uint8_t*)p = (uint8_t)v; }
// two consecutive 16-byte LDS reads issued back to back (one wait for both)
__device__ __forceinline__ void lds_load_pair16(const lz4amd_u32x4* p, lz4amd_u32x4& a, lz4amd_u32x4& b) {
    const volatile __attribute__((address_space(3))) lz4amd_u32x4* q = (const volatile __attribute__((address_space(3))) lz4amd_u32x4*)p;
    a = q[0]; b = q[1];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// four consecutive 16-byte LDS reads issued back to back (one wait for all)
__device__ __forceinline__ void lds_load_quad16(const lz4amd_u32x4* p, lz4amd_u32x4& a, lz4amd_u32x4& b, lz4amd_u32x4& c, lz4amd_u32x4& d) {
    const volatile __attribute__((address_space(3))) lz4amd_u32x4* q = (const volatile __attribute__((address_space(3))) lz4amd_u32x4*)p;
    a = q[0]; b = q[1]; c = q[2]; d = q[3];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// two control words in one trip
__device__ __forceinline__ void lds_load_2(const uint32_t* p0, const uint32_t* p1, uint32_t& v0, uint32_t& v1) {
    v0 = *(const volatile __attribute__((address_space(3))) uint32_t*)p0;
    v1 = *(const volatile __attribute__((address_space(3))) uint32_t*)p1;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// ... followed, in this order and in the same trip to the LDS, by two dwords (the DS unit serves one wave's reads in
// issue order: what the two dwords hold was published before the words in a and b were)
__device__ __forceinline__ void lds_load_pair16_then2(const lz4amd_u32x4* p, lz4amd_u32x4& a, lz4amd_u32x4& b, const uint32_t* p0, const uint32_t* p1, uint32_t& v0, uint32_t& v1) {
    const volatile __attribute__((address_space(3))) lz4amd_u32x4* q = (const volatile __attribute__((address_space(3))) lz4amd_u32x4*)p;
    a = q[0]; b = q[1];
    v0 = *(const volatile __attribute__((address_space(3))) uint32_t*)p0;
    v1 = *(const volatile __attribute__((address_space(3))) uint32_t*)p1;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// LDS operations of one wave are issued and serviced in program order; this only keeps the
// compiler from moving LDS accesses of the wave across the point (the CPU interpreter used by the
// unit tests needs a real rendezvous here, because its lanes do not run in lockstep).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// Same ordering guarantee without the wait: the DS unit executes one wave's LDS instructions in issue
// order, so a wave may queue read / write / flag-store back to back; this only stops the compiler from
// reordering them (and is a rendezvous for the CPU interpreter).
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void lds_store_relaxed(uint32_t* w, uint32_t v) {
    __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// number of set bits of a ballot below my lane (v_mbcnt)
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// x, unknown to the optimiser from here on: whatever is computed from it is computed HERE, not at the top of the kernel.
// (Everything derived from the thread index is loop invariant for the whole kernel; hoisted there, dozens of such
// values stay live through every phase of every block and push the phases' own values out to scratch memory.)
__device__ __forceinline__ uint32_t opaque_u32(uint32_t x) { asm volatile("" : "+v"(x)); return x; }
// LDS-DMA: 16 bytes per lane from global memory straight into LDS, no register in between: lane l's 16 bytes land at
// lds_dst + 16 * l (lds_dst: the same in every lane).  Asynchronous and unknown to the compiler's wait bookkeeping: the
// bytes are there after vmem_wait<N>() with N = the number of LATER memory instructions of this wave that may still be
// in flight (they complete in issue order; instructions the count misses only make the wait longer).
__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_dst) {
    const uint32_t a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(a) : "memory");
}
// all of the wave's global loads and stores complete - as an instruction the compiler's wait-count pass sees (vmcnt(0), the other counters open)
__device__ __forceinline__ void vmem_wait_all() { __builtin_amdgcn_s_waitcnt(0x0F70); }
template <int N> __device__ __forceinline__ void vmem_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }
// s_wakeup: every wave of the workgroup that sits in an s_sleep goes on at once (a wave that is not sleeping ignores it).  A wave that
// waits for another one's LDS word sleeps LONG and is woken by the writer: no polling traffic, no poll period on the hand-off's path.
__device__ __forceinline__ void wake_workgroup() { asm volatile("s_wakeup" ::: "memory"); }
#ifndef LZ4AMD_DEC_SLEEP
#define LZ4AMD_DEC_SLEEP 24
#endif
__device__ __forceinline__ void sleep_until_woken() { __builtin_amdgcn_s_sleep(LZ4AMD_DEC_SLEEP); }      // (at most ~1.5 K cycles: a wake-up that came a moment before the sleep is lost)
__device__ __forceinline__ void spin_pause() { __builtin_amdgcn_s_sleep(1); }
__device__ __forceinline__ void spin_pause_long() { __builtin_amdgcn_s_sleep(8); }
template <int N> __device__ __forceinline__ void spin_pause_n() { __builtin_amdgcn_s_sleep(N); }      // (64 * N cycles)
__device__ __forceinline__ void chain_wait_pause() { __builtin_amdgcn_s_sleep(32); }      // waiting for another workgroup
// issue priority of this wave among the waves of its SIMD (0..3)
__device__ __forceinline__ void wave_priority_high() { __builtin_amdgcn_s_setprio(3); }
// (0..3, wave-uniform; the instruction takes an immediate)
__device__ __forceinline__ void wave_priority(uint32_t p) { if (p == 0) __builtin_amdgcn_s_setprio(0); else if (p == 1) __builtin_amdgcn_s_setprio(1); else if (p == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }

// {hi,lo} >> (8 * (sh & 3)), low 32 bits (v_alignbyte_b32)
__device__ __forceinline__ uint32_t align_bytes(uint32_t hi, uint32_t lo, uint32_t sh) {
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
}
// unaligned 16-byte global accesses (gfx950 global memory takes any byte alignment)
__device__ __forceinline__ lz4amd_u32x4 ld_global16_raw(const uint8_t* p) { lz4amd_u32x4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st_global16_raw(uint8_t* p, const lz4amd_u32x4& v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ lz4amd_u32x4 ld_global16_raw(lz4amd_gsrc p) { lz4amd_u32x4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st_global16_raw(lz4amd_gdst p, const lz4amd_u32x4& v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ uint64_t ld_u64_g(lz4amd_gsrc p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ void st_global8_raw(lz4amd_gdst p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
__device__ __forceinline__ uint64_t clock_ticks() { return __builtin_readcyclecounter(); }

// value of v in lane l (l wave-uniform): v_readlane_b32, no LDS round trip
__device__ __forceinline__ uint32_t wave_readlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// ---- wave-wide scans and reductions on the DPP lanes (no LDS round trip, unlike __shfl = ds_bpermute).
// gfx9 recipe: row_shr 1,2,4,8 inside the four rows of 16 lanes, then row_bcast:15 / row_bcast:31 carry
// the row totals forward (identity 0 for lanes that receive nothing).
#define LZ4AMD_DPP(old, v, ctrl, rowmask) (uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(v), (ctrl), (rowmask), 0xf, false)
__device__ __forceinline__ uint32_t wave_incl_sum_u32(uint32_t v) {
    v += LZ4AMD_DPP(0, v, 0x111, 0xf); v += LZ4AMD_DPP(0, v, 0x112, 0xf);
    v += LZ4AMD_DPP(0, v, 0x114, 0xf); v += LZ4AMD_DPP(0, v, 0x118, 0xf);
    v += LZ4AMD_DPP(0, v, 0x142, 0xa);          // row_bcast:15 into rows 1 and 3
    v += LZ4AMD_DPP(0, v, 0x143, 0xc);          // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_max_u32(uint32_t v) {
    uint32_t t;
    t = LZ4AMD_DPP(0, v, 0x111, 0xf); v = t > v ? t : v; t = LZ4AMD_DPP(0, v, 0x112, 0xf); v = t > v ? t : v;
    t = LZ4AMD_DPP(0, v, 0x114, 0xf); v = t > v ? t : v; t = LZ4AMD_DPP(0, v, 0x118, 0xf); v = t > v ? t : v;
    t = LZ4AMD_DPP(0, v, 0x142, 0xa); v = t > v ? t : v;
    t = LZ4AMD_DPP(0, v, 0x143, 0xc); v = t > v ? t : v;
    return v;
}
// value of the lane below / above (0 for lane 0 / lane 63): wave_shr:1 / wave_shl:1 on the DPP lanes
__device__ __forceinline__ uint32_t wave_prev_u32(uint32_t v) { return LZ4AMD_DPP(0, v, 0x138, 0xf); }
__device__ __forceinline__ uint32_t wave_next_u32(uint32_t v) { return LZ4AMD_DPP(0, v, 0x130, 0xf); }
// minimum over each row of 16 lanes, in every lane of the row (row_ror 8,4,2,1)
__device__ __forceinline__ uint32_t row16_min_u32(uint32_t v) {
    uint32_t t;
    t = LZ4AMD_DPP(v, v, 0x128, 0xf); v = t < v ? t : v; t = LZ4AMD_DPP(v, v, 0x124, 0xf); v = t < v ? t : v;
    t = LZ4AMD_DPP(v, v, 0x122, 0xf); v = t < v ? t : v; t = LZ4AMD_DPP(v, v, 0x121, 0xf); v = t < v ? t : v;
    return v;
}

// device-scope work-queue ticket
// A word another WORKGROUP publishes (linked blocks: where the predecessor's output ended).  Agent scope: the writer's
// earlier stores are visible to this CU's later loads once the value is seen (release -> acquire, MI355X_MICROARCH.md
// "Correctness boundaries": workgroup scope is not enough across CUs / XCDs).
__device__ __forceinline__ long long chain_load_acquire(const long long* w) {
    return __hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_store_release(long long* w, long long v) {
    __hip_atomic_store(w, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t flag_load_agent(const uint32_t* w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flag_store_agent(uint32_t* w, uint32_t v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t take_ticket(uint32_t* counter) {
    return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the same hand-offs for kernels whose waves exchange nothing but LDS data (the fast compressor): no wait for the wave's global
// loads and stores.  (A release / acquire at workgroup scope, and __builtin_amdgcn_fence, cover every address space: they wait for the
// acknowledgement of global stores nobody in the workgroup reads, and for prefetches that are wanted a tile later.)  The DS unit serves
// one wave's LDS instructions in issue order: a flag stored after the data is seen after the data; the reader's data reads are issued
// after the flag's value came back.  What is left to do is to keep the compiler from reordering.
__device__ __forceinline__ void wave_lds_fence_local() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ void lds_store_release_local(uint32_t* w, uint32_t v) {
    asm volatile("" ::: "memory"); __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_or_release_local(uint32_t* w, uint32_t bits) {
    asm volatile("" ::: "memory"); __hip_atomic_fetch_or(w, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("" ::: "memory");
}
__device__ __forceinline__ uint32_t lds_load_acquire_local(const uint32_t* w) {
    asm volatile("" ::: "memory"); const uint32_t v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return v;
}
// Workgroup barrier for data that travels through LDS only: the wave's LDS operations are complete (lgkmcnt), its global loads and
// stores may still be in flight - __syncthreads() also waits for those (its release fence covers every address space), i.e. for the
// acknowledgement of stores nobody in the workgroup reads and for prefetches that are only wanted a tile later.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// the registers of a global load, complete from here on (the compiler waits for the load at this point, not at a later use behind
// stores whose number it cannot know: the memory counter is in order)
__device__ __forceinline__ void settle_load16(lz4amd_u32x4& v) { asm volatile("" : "+v"(v)); }
// ... the same as a mere use of the registers (no new value: the compiler keeps them where they are)
__device__ __forceinline__ void touch_load16(const lz4amd_u32x4& v) { asm volatile("" :: "v"(v)); }

// lanes of a wave run in lockstep on the hardware; this only pins the compiler's schedule
// (and gives the CPU interpreter used by the unit tests a rendezvous point).
__device__ __forceinline__ void wave_converge() { __builtin_amdgcn_wave_barrier(); }
