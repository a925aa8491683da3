// micro-test: do unaligned LDS accesses / LDS-DMA work on gfx950, and what do they cost?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <stdlib.h>
#include <string.h>
struct alignas(16) U4 { uint32_t x,y,z,w; };
extern __shared__ __attribute__((aligned(16))) char smem[];
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_fill_read(const uint32_t* idx, U4* out, uint64_t* out8, uint32_t* out4) {
    for (uint32_t i = threadIdx.x; i < 65536; i += blockDim.x) smem[i] = (char)(i * 7 + (i >> 8));
    __syncthreads();
    uint32_t a = idx[threadIdx.x];
    U4 v; __builtin_memcpy(&v, smem + a, 16); out[threadIdx.x] = v;
    uint64_t v8; __builtin_memcpy(&v8, smem + a + 1, 8); out8[threadIdx.x] = v8;
    uint32_t v4; __builtin_memcpy(&v4, smem + a + 2, 4); out4[threadIdx.x] = v4;
}
__global__ void k_write(const uint32_t* idx, uint8_t* out) {
    for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) smem[i] = 0;
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t a = threadIdx.x * 48 + (idx[threadIdx.x] & 15);
        U4 v; v.x = 0x03020100u + threadIdx.x; v.y = 0x07060504u; v.z = 0x0b0a0908u; v.w = 0x0f0e0d0cu;
        __builtin_memcpy(smem + a, &v, 16);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) out[i] = smem[i];
}
__global__ void k_dma(const uint8_t* g, uint8_t* out) {
    // 256 threads: each wave DMAs 1 KB from g + wave*1024 (+ lane*16) into smem + wave*1024
    uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + w * 1024 + l * 16),
        (__attribute__((address_space(3))) void*)(smem + w * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) out[i] = smem[i];
}
template <int MODE>
__global__ void k_bench(const uint32_t* idx, uint32_t* out, int iters) {
    for (uint32_t i = threadIdx.x; i < 65536 + 64; i += blockDim.x) smem[i] = (char)i;
    __syncthreads();
    uint32_t a = idx[threadIdx.x];
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { U4 v; __builtin_memcpy(&v, smem + a, 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
        else if (MODE == 1) {
            const uint32_t* r = (const uint32_t*)(smem + (a & ~3u)); uint32_t sh = a & 3;
            uint32_t d0 = r[0], d1 = r[1], d2 = r[2], d3 = r[3], d4 = r[4];
            acc += __builtin_amdgcn_alignbyte(d1, d0, sh) ^ __builtin_amdgcn_alignbyte(d2, d1, sh) ^ __builtin_amdgcn_alignbyte(d3, d2, sh) ^ __builtin_amdgcn_alignbyte(d4, d3, sh);
        } else { U4 v = *(const U4*)(smem + (a & ~15u)); acc += v.x ^ v.y ^ v.z ^ v.w; }
        a = (a + 4099 + (acc & 3)) & 0xFFFF;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    const int T = 1024;
    std::vector<uint32_t> idx(T);
    srand(1); for (int i = 0; i < T; i++) idx[i] = rand() % 60000;
    uint32_t* d_idx; U4* d_out; uint64_t* d_o8; uint32_t* d_o4; uint8_t* d_b; uint8_t* d_g;
    CHK(hipMalloc(&d_idx, T * 4)); CHK(hipMalloc(&d_out, T * 16)); CHK(hipMalloc(&d_o8, T * 8)); CHK(hipMalloc(&d_o4, T * 4));
    CHK(hipMalloc(&d_b, 4096)); CHK(hipMalloc(&d_g, 8192));
    CHK(hipMemcpy(d_idx, idx.data(), T * 4, hipMemcpyHostToDevice));
    CHK(hipFuncSetAttribute((const void*)k_fill_read, hipFuncAttributeMaxDynamicSharedMemorySize, 70000));
    k_fill_read<<<1, T, 70000>>>(d_idx, d_out, d_o8, d_o4);
    CHK(hipDeviceSynchronize());
    std::vector<U4> out(T); std::vector<uint64_t> o8(T); std::vector<uint32_t> o4(T);
    CHK(hipMemcpy(out.data(), d_out, T * 16, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(o8.data(), d_o8, T * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(o4.data(), d_o4, T * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    auto B = [](uint32_t i) { return (uint8_t)(i * 7 + (i >> 8)); };
    for (int t = 0; t < T; t++) {
        uint8_t e[16]; for (int k = 0; k < 16; k++) e[k] = B(idx[t] + k);
        if (memcmp(e, &out[t], 16)) bad++;
        uint8_t e8[8]; for (int k = 0; k < 8; k++) e8[k] = B(idx[t] + 1 + k);
        if (memcmp(e8, &o8[t], 8)) bad++;
        uint8_t e4[4]; for (int k = 0; k < 4; k++) e4[k] = B(idx[t] + 2 + k);
        if (memcmp(e4, &o4[t], 4)) bad++;
    }
    printf("unaligned LDS reads b128/b64/b32: %s (%d bad)\n", bad ? "FAIL" : "ok", bad);
    k_write<<<1, 256, 4096>>>(d_idx, d_b);
    CHK(hipDeviceSynchronize());
    std::vector<uint8_t> hb(4096); CHK(hipMemcpy(hb.data(), d_b, 4096, hipMemcpyDeviceToHost));
    bad = 0;
    for (int t = 0; t < 64; t++) { uint32_t a = t * 48 + (idx[t] & 15); for (int k = 0; k < 16; k++) { uint8_t e = (k == 0) ? (uint8_t)(0 + t) : (uint8_t)k; if (k < 4) { uint32_t w = 0x03020100u + t; e = (uint8_t)(w >> (8 * k)); } if (hb[a + k] != e) bad++; } }
    printf("unaligned LDS write b128: %s (%d bad)\n", bad ? "FAIL" : "ok", bad);
    std::vector<uint8_t> g(8192); for (int i = 0; i < 8192; i++) g[i] = (uint8_t)(i * 13 + 5);
    CHK(hipMemcpy(d_g, g.data(), 8192, hipMemcpyHostToDevice));
    for (int skew = 0; skew <= 16; skew += 4) {
        k_dma<<<1, 256, 4096>>>(d_g + skew, d_b);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("dma skew %d: error %s\n", skew, hipGetErrorString(e)); break; }
        CHK(hipMemcpy(hb.data(), d_b, 4096, hipMemcpyDeviceToHost));
        bad = 0; for (int i = 0; i < 4096; i++) if (hb[i] != g[i + skew]) bad++;
        printf("LDS-DMA dwordx4, global skew %d: %s (%d bad)\n", skew, bad ? "FAIL" : "ok", bad);
    }
    // throughput
    uint32_t* d_acc; CHK(hipMalloc(&d_acc, 256 * T * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CHK(hipEventRecord(e0));
            if (mode == 0) { CHK(hipFuncSetAttribute((const void*)k_bench<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 70000)); k_bench<0><<<256, T, 70000>>>(d_idx, d_acc, iters); }
            if (mode == 1) { CHK(hipFuncSetAttribute((const void*)k_bench<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 70000)); k_bench<1><<<256, T, 70000>>>(d_idx, d_acc, iters); }
            if (mode == 2) { CHK(hipFuncSetAttribute((const void*)k_bench<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 70000)); k_bench<2><<<256, T, 70000>>>(d_idx, d_acc, iters); }
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("mode %d (%s): %.3f ms -> %.1f cycles per wave-read per CU (16 waves, dependent chain)\n", mode,
                mode == 0 ? "unaligned b128" : mode == 1 ? "5xb32+alignbyte" : "aligned b128", ms, ms * 1e-3 * 2.4e9 / iters / 16);
        }
    }
    return 0;
}
