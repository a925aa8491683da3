// platform_emu.h -- TEST INFRASTRUCTURE ONLY: CPU-interpreter twin of
// lz4_amd/csrc/kernels/platform_hip.h (see simt_emu.h).
#pragma once
#include "simt_emu.h"
#include <stdint.h>
static inline void lds_or_release(uint32_t* w, uint32_t bits) { *w |= bits; }
static inline uint32_t lds_load_acquire(const uint32_t* w) { return *(volatile const uint32_t*)w; }
static inline void spin_pause() {}
static inline uint32_t take_ticket(uint32_t* counter) { return __atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED); }
static inline void wave_converge() { (void)__ballot(1); }
