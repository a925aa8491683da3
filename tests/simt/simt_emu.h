/*
 * simt_emu.h -- TEST INFRASTRUCTURE ONLY: a host-side SIMT interpreter for the device
 * code under lz4_amd/csrc/kernels/.
 *
 * There is no GPU in the build container, and GPU minutes on the MI355X box are scarce,
 * so the kernels' LOGIC is exercised on the CPU by compiling the very same kernel bodies
 * (the *.h files under csrc/kernels are plain C++ with HIP builtins) against this header
 * instead of <hip/hip_runtime.h>.  Every "thread" of a workgroup is a user-mode fiber; a
 * wave is 64 consecutive fibers; wave-wide builtins (__ballot, __shfl, ...) and
 * __syncthreads() are rendezvous points between fibers.  This is NOT a product path: the
 * shipped library contains only the hipcc-compiled gfx950 kernels and fails loudly without
 * a GPU.  Nothing outside tests/ includes this file.
 *
 * Restrictions (kernels are written to satisfy them; they are also good CDNA practice):
 *   - wave-wide builtins are called in wave-uniform control flow (all live lanes arrive);
 *   - LDS is only the dynamic region (LZ4AMD_DYN_LDS) -- poisoned per block here;
 *   - blockDim.x is a multiple of 64, 1-D grids/blocks only.
 * x86-64 SysV only (hand-written context switch).
 */
#ifndef SIMT_EMU_H
#define SIMT_EMU_H

#include <cstdint>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <functional>
#include <thread>
#include <atomic>

namespace simt {

struct dim3_t { unsigned x, y, z; };

struct Wave;
struct Block;

struct Fiber {
    void* sp;               // saved stack pointer
    char* stack;
    unsigned tid;
    bool done;
    unsigned gen;           // wave-op generation this lane is at
    unsigned bar_gen;       // block-barrier generation this thread is at
    int waiting;            // 0 running, 1 in a wave op, 2 in a block barrier (diagnostics)
    Wave* wave;
    Block* block;
};

// A rendezvous counter tagged with the generation it is counting for.  Parity p serves
// generations g, g+2, ...; nobody can arrive at g+2 before everybody has left g (leaving
// g+1's rendezvous requires every live participant to have arrived there), so the first
// arriver of a new generation may safely reset the counter.
struct Rendezvous {
    unsigned gen[2];
    unsigned count[2];
    unsigned acc[2];
};

struct Wave {
    uint64_t slot[2][64];   // deposited arguments, double-buffered by op parity
    Rendezvous rv;
    uint64_t present[2];    // lanes that deposited for the op of this parity
    unsigned live;
    unsigned first;         // tid of lane 0
};

struct Block {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    unsigned nthreads;
    Rendezvous bar;
    unsigned live_threads;
    char* smem;
    dim3_t bIdx, bDim, gDim;
    void* sched_sp;
    Fiber* cur;
    const std::function<void()>* body;
    unsigned long long progress;
};

extern thread_local Block* g_blk;

extern "C" void simt_switch(void** save_sp, void* load_sp);

inline Fiber* cur() { return g_blk->cur; }

inline void yield_to_sched() {
    Block* b = g_blk;
    Fiber* f = b->cur;
    simt_switch(&f->sp, b->sched_sp);
}
// waiting for ANOTHER block (running on another host thread): not a deadlock of this block's lanes
inline void external_wait() { g_blk->progress++; sched_yield(); yield_to_sched(); }


// arrive at generation g of rendezvous rv and wait until `*live` participants have arrived
inline unsigned rendezvous(Rendezvous& rv, unsigned g, const unsigned* live, unsigned contrib, int mode) {
    unsigned p = g & 1;
    if (rv.gen[p] != g + 1) { rv.gen[p] = g + 1; rv.count[p] = 0; rv.acc[p] = 0; }
    rv.count[p]++;
    if (mode == 1) rv.acc[p] |= (contrib != 0);
    if (mode == 2) rv.acc[p] += (contrib != 0);
    g_blk->cur->waiting = (live == &g_blk->live_threads) ? 2 : 1;
    while (rv.count[p] < *live) yield_to_sched();
    g_blk->cur->waiting = 0;
    g_blk->progress++;
    return rv.acc[p];
}

// ---- wave rendezvous: deposit v, wait for all live lanes, then let fn read the slots
template <class F>
inline auto wave_op(uint64_t v, F&& fn) -> decltype(fn((const uint64_t*)nullptr, 0u)) {
    Fiber* f = cur();
    Wave* w = f->wave;
    unsigned par = f->gen & 1;
    unsigned lane = f->tid - w->first;
    if (w->rv.gen[par] != f->gen + 1) w->present[par] = 0;     // first arriver of this op
    w->slot[par][lane] = v;
    w->present[par] |= 1ull << lane;
    rendezvous(w->rv, f->gen, &w->live, 0, 0);
    auto r = fn(w->slot[par], lane);
    f->gen++;
    return r;
}

inline unsigned block_barrier(unsigned contrib, int mode /*0 none,1 or,2 count*/) {
    Block* b = g_blk;
    Fiber* f = b->cur;
    unsigned r = rendezvous(b->bar, f->bar_gen, &b->live_threads, contrib, mode);
    if (getenv("SIMT_DEBUG")) fprintf(stderr, "bar tid %u gen %u mode %d contrib %u -> %u\n", f->tid, f->bar_gen, mode, contrib, r);
    f->bar_gen++;
    return r;
}

void launch(unsigned grid, unsigned block, size_t smem_bytes, const std::function<void()>& body,
            size_t stack_bytes = 128 << 10);

} // namespace simt

// ------------------------------------------------------------------ HIP surface
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __noinline__ __attribute__((noinline))

#define threadIdx (simt::dim3_t{simt::cur()->tid, 0, 0})
#define blockIdx  (simt::g_blk->bIdx)
#define blockDim  (simt::g_blk->bDim)
#define gridDim   (simt::g_blk->gDim)

// dynamic LDS region (the HIP build defines this as `extern __shared__ ... char name[]`)
#define LZ4AMD_DYN_LDS(name) char* name = simt::g_blk->smem

static inline void __syncthreads() { simt::block_barrier(0, 0); }
static inline int __syncthreads_or(int p) { return (int)simt::block_barrier((unsigned)p, 1) != 0; }
static inline int __syncthreads_count(int p) { return (int)simt::block_barrier((unsigned)p, 2); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

static inline unsigned long long __ballot(int pred) {
    return simt::wave_op((uint64_t)(pred != 0), [](const uint64_t* s, unsigned) {
        unsigned long long m = 0;
        simt::Fiber* f = simt::cur();
        uint64_t present = f->wave->present[f->gen & 1];
        for (unsigned i = 0; i < 64; i++)
            if (((present >> i) & 1) && s[i]) m |= 1ull << i;
        return m;
    });
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) { return __ballot(!pred) == 0; }

template <class T> static inline T __shfl(T v, int srcLane) {
    static_assert(sizeof(T) <= 8, "shfl of <=8 byte types only");
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
    // two-step: everyone deposits value; each lane reads slot[srcLane]
    uint64_t got = simt::wave_op(raw, [srcLane](const uint64_t* s, unsigned) { return s[srcLane & 63]; });
    T out; memcpy(&out, &got, sizeof(T)); return out;
}
template <class T> static inline T __shfl_up(T v, unsigned d) {
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
    uint64_t got = simt::wave_op(raw, [d](const uint64_t* s, unsigned lane) { return lane >= d ? s[lane - d] : s[lane]; });
    T out; memcpy(&out, &got, sizeof(T)); return out;
}
template <class T> static inline T __shfl_down(T v, unsigned d) {
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
    uint64_t got = simt::wave_op(raw, [d](const uint64_t* s, unsigned lane) { return lane + d < 64 ? s[lane + d] : s[lane]; });
    T out; memcpy(&out, &got, sizeof(T)); return out;
}
template <class T> static inline T __shfl_xor(T v, int m) {
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
    uint64_t got = simt::wave_op(raw, [m](const uint64_t* s, unsigned lane) { return s[(lane ^ (unsigned)m) & 63]; });
    T out; memcpy(&out, &got, sizeof(T)); return out;
}
static inline unsigned __builtin_amdgcn_readfirstlane(unsigned v) {
    return (unsigned)simt::wave_op((uint64_t)v, [](const uint64_t* s, unsigned) {
        simt::Fiber* f = simt::cur();
        uint64_t present = f->wave->present[f->gen & 1];
        return present ? s[__builtin_ctzll(present)] : (uint64_t)0;
    });
}
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline int __umul24(unsigned a, unsigned b) { return (int)((a & 0xFFFFFFu) * (b & 0xFFFFFFu)); }      // (HIP declares it returning int - amd_device_functions.h: an expression of such products shifts arithmetically unless it goes through an unsigned first)

// atomics: fibers of one block never run concurrently; blocks on different OS threads may,
// so global atomics use real atomics (harmless for LDS).
template <class T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicSub(T* p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicOr(T* p, T v)  { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicAnd(T* p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }

#endif
