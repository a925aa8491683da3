// platform_hip.h -- the gfx950 build's view of the few primitives the kernel bodies use
// beyond plain HIP builtins.  (tests/simt/platform_emu.h is the CPU-interpreter twin used
// only by the unit tests; the shipped library is built from this file alone.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// All LDS lives in the dynamic region, 16-byte aligned (CDNA guide G17).
#define LZ4AMD_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]

// workgroup-scope release/acquire on LDS words (waves of one workgroup hand data to each
// other through LDS without a barrier; LDS is coherent inside a workgroup).
__device__ __forceinline__ void lds_or_release(uint32_t* w, uint32_t bits) {
    __hip_atomic_fetch_or(w, bits, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t lds_load_acquire(const uint32_t* w) {
    return __hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void spin_pause() { __builtin_amdgcn_s_sleep(2); }

// device-scope work-queue ticket
__device__ __forceinline__ uint32_t take_ticket(uint32_t* counter) {
    return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// lanes of a wave run in lockstep on the hardware; this only pins the compiler's schedule
// (and gives the CPU interpreter used by the unit tests a rendezvous point).
__device__ __forceinline__ void wave_converge() { __builtin_amdgcn_wave_barrier(); }
