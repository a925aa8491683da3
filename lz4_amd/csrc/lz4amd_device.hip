// lz4amd_device.hip -- the only HIP translation unit of liblz4_amd: gfx950 kernels plus the thin
// extern "C" launch / memory wrappers declared in lz4amd_ffi.h.  Kernel bodies live in kernels/*.h.
#include "kernels/platform_hip.h"
#include "kernels/lz4_decompress_kernel.h"
#include "kernels/lz4_compress_kernel.h"
#include "kernels/lz4_hc_kernel.h"
#include "kernels/xxh32_kernel.h"
#include "kernels/gather_kernel.h"
#include "kernels/chain_spec_kernel.h"
#include "lz4amd_ffi.h"
#include <stdio.h>
#include <stdlib.h>

using namespace lz4amd;

// ------------------------------------------------------------------------------- kernels
__global__ void __launch_bounds__(kDecThreads) lz4amd_k_decompress(lz4amd_dec_params p) { decompress_batch_body_of<false>(p); }
__global__ void __launch_bounds__(kDecThreads) lz4amd_k_decompress_runs(lz4amd_dec_params p) { decompress_batch_body_of<true>(p); }      // dependent blocks (p.chain)
__global__ void __launch_bounds__(kCmpThreads) lz4amd_k_compress(lz4amd_comp_params p) { compress_batch_body(p); }

__global__ void __launch_bounds__(kHcThreads) lz4amd_k_compress_hc(lz4amd_hc_params p) { hc_batch_body(p); }
__global__ void __launch_bounds__(64) lz4amd_k_xxh32(lz4amd_xxh_params p) { xxh32_block_body(p); }
__global__ void __launch_bounds__(256) lz4amd_k_gather(lz4amd_gather_params p) { gather_block_body(p); }

__global__ void __launch_bounds__(kSpecThreads) lz4amd_k_spec_fill(lz4amd_spec_params p) { spec_fill_body(p); }
__global__ void __launch_bounds__(kSpecScanThreads) lz4amd_k_spec_scan(lz4amd_spec_params p) { spec_scan_body(p); }
__global__ void __launch_bounds__(kSpecThreads) lz4amd_k_spec_gate(lz4amd_spec_params p) { spec_gate_body(p); }
__global__ void __launch_bounds__(kSpecThreads) lz4amd_k_spec_merge(lz4amd_spec_params p) { spec_merge_body(p); }
__global__ void __launch_bounds__(kSpecPatchThreads) lz4amd_k_spec_patch(lz4amd_spec_params p) { spec_patch_body(p); }
__global__ void __launch_bounds__(kSpecScanThreads) lz4amd_k_spec_results(lz4amd_spec_params p) { spec_results_body(p); }

// calibration: a plain 16-bytes-per-lane stream copy, the bandwidth this box's HBM actually delivers to a read+write stream.
// U granules per thread and trip, all loads issued before the first store (U * 16 bytes in flight per lane); NT: non-temporal stores.
template <int U, bool NT>
__global__ void __launch_bounds__(256) lz4amd_k_stream_copy(const lz4amd_u32x4* __restrict__ src, lz4amd_u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * U;
    size_t i = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x;
    for (; i + (size_t)(U - 1) * blockDim.x < n16; i += stride) {
        lz4amd_u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(&src[i + (size_t)k * blockDim.x]) : src[i + (size_t)k * blockDim.x];
#pragma unroll
        for (int k = 0; k < U; k++) { if (NT) __builtin_nontemporal_store(v[k], &dst[i + (size_t)k * blockDim.x]); else dst[i + (size_t)k * blockDim.x] = v[k]; }
    }
    for (int k = 0; k < U; k++) { const size_t j = i + (size_t)k * blockDim.x; if (j < n16) dst[j] = src[j]; }
}

// ------------------------------------------------------------------------------- runtime glue
static thread_local char g_err[256] = "";
static int fail(hipError_t e, const char* what) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return -1;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(e_, #call); } while (0)

extern "C" const char* lz4amd_hip_errstr(void) { return g_err; }

extern "C" int lz4amd_hip_init(int device, int* n_cus) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        snprintf(g_err, sizeof g_err, "no HIP device (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
        return -1;
    }
    if (device < 0 || device >= count) { snprintf(g_err, sizeof g_err, "device %d out of range (%d)", device, count); return -1; }
    HIPCHK(hipSetDevice(device));
    int cus = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    if (n_cus) *n_cus = cus;
    // the decoder uses ~152 KB of the CU's 160 KB LDS: opt in to large dynamic LDS
    HIPCHK(hipFuncSetAttribute((const void*)lz4amd_k_decompress, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDecLdsBytes));
    HIPCHK(hipFuncSetAttribute((const void*)lz4amd_k_decompress_runs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDecLdsBytes));
    HIPCHK(hipFuncSetAttribute((const void*)lz4amd_k_compress, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCmpLdsBytes));
    HIPCHK(hipFuncSetAttribute((const void*)lz4amd_k_compress_hc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHcLdsBytes));
    HIPCHK(hipFuncSetAttribute((const void*)lz4amd_k_spec_patch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSpecPatchLds));
    return 0;
}

// Every entry point that takes a context or a plan selects the context's device first: contexts on different GPUs, or a
// context used from another thread than the one that made it, must not land on whatever device is current there.
extern "C" int lz4amd_hip_use_device(int device) { HIPCHK(hipSetDevice(device)); return 0; }

extern "C" void* lz4amd_hip_malloc(size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) { fail(e, "hipMalloc"); return nullptr; }
    return p;
}
extern "C" void lz4amd_hip_free(void* d) { if (d) (void)hipFree(d); }
extern "C" int lz4amd_hip_h2d(void* d, const void* h, size_t n, void* s) {
    if (!n) return 0;
    HIPCHK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)s)); return 0;
}
extern "C" int lz4amd_hip_d2h(void* h, const void* d, size_t n, void* s) {
    if (!n) return 0;
    HIPCHK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, (hipStream_t)s)); return 0;
}
extern "C" int lz4amd_hip_memset(void* d, int v, size_t n, void* s) {
    if (!n) return 0;
    HIPCHK(hipMemsetAsync(d, v, n, (hipStream_t)s)); return 0;
}
extern "C" int lz4amd_hip_sync(void* s) { HIPCHK(hipStreamSynchronize((hipStream_t)s)); return 0; }
// page-locked host memory, readable and writable by the device through the same pointer
extern "C" void* lz4amd_hip_host_alloc(size_t bytes) { void* p = nullptr; if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr; return p; }
extern "C" void lz4amd_hip_host_free(void* p) { if (p) (void)hipHostFree(p); }
extern "C" void* lz4amd_hip_stream_create(void) { hipStream_t s; if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr; return (void*)s; }
extern "C" void lz4amd_hip_stream_destroy(void* s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }
extern "C" void* lz4amd_hip_event_create(void) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; return (void*)e; }
extern "C" void lz4amd_hip_event_destroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }
extern "C" int lz4amd_hip_event_record(void* ev, void* s) { HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)s)); return 0; }
extern "C" int lz4amd_hip_event_sync(void* ev) { HIPCHK(hipEventSynchronize((hipEvent_t)ev)); return 0; }
extern "C" float lz4amd_hip_event_ms(void* a, void* b) {
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) return -1.f;
    return ms;
}

extern "C" size_t lz4amd_hip_dec_scratch_bytes(unsigned max_csize, unsigned max_out) { return (size_t)dec_scratch_bytes(max_csize, max_out); }
extern "C" size_t lz4amd_hip_hc_scratch_bytes(unsigned max_src) { return (size_t)hc_scratch_bytes(max_src); }
extern "C" int lz4amd_hip_launch_compress_hc(const lz4amd_hc_params* p, unsigned grid, void* s) {
    if (!p->n_blocks || !grid) return 0;
    HIPCHK(hipMemsetAsync(p->ticket, 0, sizeof(uint32_t), (hipStream_t)s));
    hipLaunchKernelGGL(lz4amd_k_compress_hc, dim3(grid), dim3(kHcThreads), kHcLdsBytes, (hipStream_t)s, *p);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int lz4amd_hip_launch_stream_copy(void* d_dst, const void* d_src, size_t bytes, unsigned grid, unsigned variant, void* s) {
    if (bytes < 16 || !grid) return 0;
    const lz4amd_u32x4* a = (const lz4amd_u32x4*)d_src; lz4amd_u32x4* b = (lz4amd_u32x4*)d_dst;
    switch (variant) {
    case 0: hipLaunchKernelGGL((lz4amd_k_stream_copy<1, false>), dim3(grid), dim3(256), 0, (hipStream_t)s, a, b, bytes / 16); break;
    case 1: hipLaunchKernelGGL((lz4amd_k_stream_copy<4, false>), dim3(grid), dim3(256), 0, (hipStream_t)s, a, b, bytes / 16); break;
    case 2: hipLaunchKernelGGL((lz4amd_k_stream_copy<4, true>), dim3(grid), dim3(256), 0, (hipStream_t)s, a, b, bytes / 16); break;
    case 3: hipLaunchKernelGGL((lz4amd_k_stream_copy<8, false>), dim3(grid), dim3(256), 0, (hipStream_t)s, a, b, bytes / 16); break;
    default: hipLaunchKernelGGL((lz4amd_k_stream_copy<2, true>), dim3(grid), dim3(256), 0, (hipStream_t)s, a, b, bytes / 16); break;
    }
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int lz4amd_hip_launch_xxh32(const lz4amd_xxh_params* p, void* s) {
    if (!p->n_blocks) return 0;
    hipLaunchKernelGGL(lz4amd_k_xxh32, dim3(p->n_blocks), dim3(64), kXxhChunk, (hipStream_t)s, *p);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int lz4amd_hip_launch_gather(const lz4amd_gather_params* p, void* s) {
    if (!p->n_blocks) return 0;
    hipLaunchKernelGGL(lz4amd_k_gather, dim3(p->n_blocks * kGatherSlices), dim3(kGatherThreads), 0, (hipStream_t)s, *p);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int lz4amd_hip_launch_decompress(const lz4amd_dec_params* p, unsigned grid, void* s) {
    if (!p->n_blocks || !grid) return 0;
    HIPCHK(hipMemsetAsync(p->ticket, 0, sizeof(uint32_t), (hipStream_t)s));
    if (p->chain) {                      /* dependent blocks: nothing is known but where the first one starts */
        HIPCHK(hipMemsetAsync(p->chain, 0xFF, LZ4AMD_CHAIN_RESET_BYTES(p->n_blocks), (hipStream_t)s));     /* (the words, what the blocks report and carry: "not known yet") */
        HIPCHK(hipMemsetAsync(p->chain, 0, sizeof(long long), (hipStream_t)s));
    }
    if (p->chain) hipLaunchKernelGGL(lz4amd_k_decompress_runs, dim3(grid), dim3(kDecThreads), kDecLdsBytes, (hipStream_t)s, *p);
    else hipLaunchKernelGGL(lz4amd_k_decompress, dim3(grid), dim3(kDecThreads), kDecLdsBytes, (hipStream_t)s, *p);
    HIPCHK(hipGetLastError());
    return 0;
}
// linked blocks side by side (kernels/chain_spec_kernel.h): the made-up histories once, when the plan is made ...
extern "C" int lz4amd_hip_launch_spec_fill(const lz4amd_spec_params* p, void* s) {
    if (p->n_units < 2) return 0;
    hipLaunchKernelGGL(lz4amd_k_spec_fill, dim3(3 * (p->n_units - 1) * kSpecFillParts), dim3(kSpecThreads), 0, (hipStream_t)s, *p);
    HIPCHK(hipGetLastError());
    return 0;
}
// ... and per launch: the decoder over unit 0 and variants A, B (C where it is needed) of every other unit, positions, merge, patch, results
extern "C" int lz4amd_hip_launch_spec(const lz4amd_spec_params* p, const lz4amd_dec_params* dec, unsigned dec_grid,
                                      const lz4amd_dec_params* dec_b, unsigned dec_b_grid, unsigned max_cap, void* s) {
    if (!p->n) return 0;
    if (lz4amd_hip_launch_decompress(dec, dec_grid, s)) return -1;
    if (dec_b) {                          // chains of large blocks: the second (and where needed third) copies, from the tables the first launch wrote
        hipLaunchKernelGGL(lz4amd_k_spec_gate, dim3(p->n_units - 1), dim3(kSpecThreads), 0, (hipStream_t)s, *p);
        if (lz4amd_hip_launch_decompress(dec_b, dec_b_grid, s)) return -1;
    }
    const unsigned slices = (max_cap + kSpecSlice - 1) / kSpecSlice;
    hipLaunchKernelGGL(lz4amd_k_spec_scan, dim3(1), dim3(kSpecScanThreads), 0, (hipStream_t)s, *p);
    hipLaunchKernelGGL(lz4amd_k_spec_merge, dim3(p->n_units, slices ? slices : 1), dim3(kSpecThreads), 0, (hipStream_t)s, *p);
    hipLaunchKernelGGL(lz4amd_k_spec_patch, dim3(p->n_units < 512 ? p->n_units : 512), dim3(kSpecPatchThreads), kSpecPatchLds, (hipStream_t)s, *p);
    hipLaunchKernelGGL(lz4amd_k_spec_results, dim3(1), dim3(kSpecScanThreads), 0, (hipStream_t)s, *p);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int lz4amd_hip_launch_compress(const lz4amd_comp_params* p, unsigned grid, void* s) {
    if (!p->n_blocks || !grid) return 0;
    HIPCHK(hipMemsetAsync(p->ticket, 0, sizeof(uint32_t), (hipStream_t)s));
    hipLaunchKernelGGL(lz4amd_k_compress, dim3(grid), dim3(kCmpThreads), kCmpLdsBytes, (hipStream_t)s, *p);
    HIPCHK(hipGetLastError());
    return 0;
}
