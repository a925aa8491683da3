// throw-away measurement: does global_load_lds_dwordx4 (LDS-DMA) land lane-linear at M0, also above 64 KB of LDS? latency?
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_dma tools/exp/lds_dma.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ void lds_dma16(const void* gsrc, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}
extern __shared__ __attribute__((aligned(16))) char smem[];
__global__ void k(const uint8_t* src, uint8_t* out, uint32_t lds_off, uint64_t* cyc) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) ((uint32_t*)(smem + lds_off))[i] = 0xDEADBEEF;
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint64_t t0 = __builtin_readcyclecounter();
        for (int c = 0; c < 8; c++) lds_dma16(src + 1024 * c + 16 * lane + 3 * 16 * 0, __builtin_amdgcn_readfirstlane(base + lds_off + 1024 * c));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t t1 = __builtin_readcyclecounter();
        if (lane == 0) cyc[0] = t1 - t0;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 8192; i += blockDim.x) out[i] = (uint8_t)smem[lds_off + i];
}
int main() {
    std::vector<uint8_t> h(16384); for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(i * 7 + (i >> 8));
    uint8_t *d, *o; uint64_t* c; hipMalloc(&d, 16384); hipMalloc(&o, 8192); hipMalloc(&c, 8);
    hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (uint32_t off : {1024u, 60u * 1024, 100u * 1024, 150u * 1024}) {
        hipMemset(o, 0, 8192);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 160 * 1024 - 256, 0, d, o, off, c);
        std::vector<uint8_t> r(8192); uint64_t cy; hipMemcpy(r.data(), o, 8192, hipMemcpyDeviceToHost); hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < 8192; i++) bad += r[i] != h[i];
        printf("lds offset %6u: %zu of 8192 bytes differ, 8 x 1 KB issue->landed %llu cycles (%s)\n", off, bad, (unsigned long long)cy, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
