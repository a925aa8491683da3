// micro-benchmark (throw-away measurement program, not part of the product): what do the LDS / VMEM
// primitives the codec kernels could be built from cost on gfx950, with a full CU (16 waves) issuing them?
// Prints cycles per wave-instruction per CU (aggregate issue interval) for INDEPENDENT operations.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
extern __shared__ __attribute__((aligned(16))) char smem[];
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define LDS(T, a) (*(volatile __attribute__((address_space(3))) T*)(uintptr_t)(uint32_t)(a))

enum { RD32 = 0, RD64, RD128, RDU8, WR8, WR32, WR64, WR128, BPERM, ATOMMAX, OR64, RD32X2, GST16, GST1, GST4, NMODES };
static const char* names[] = { "ds_read_b32", "ds_read_b64", "ds_read_b128", "ds_read_u8", "ds_write_b8", "ds_write_b32", "ds_write_b64", "ds_write_b128",
                               "ds_bpermute_b32", "ds_max_u32 (no rtn)", "ds_or_b64 (no rtn)", "2 x ds_read_b32 (a, a+4)", "global_store_b128", "global_store_b8", "global_store_b32" };

// addressing patterns: lane stride (bytes), misalignment (bytes), random (per-lane random base)
template <int MODE>
__global__ void __launch_bounds__(1024) k(uint32_t stride, uint32_t mis, uint32_t rnd, int iters, uint32_t* out, uint8_t* gbuf) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (uint32_t i = tid; i < 65536 / 4; i += 1024) ((uint32_t*)smem)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a = w * 4096 + lane * stride + mis;
    uint32_t h = (tid * 2654435761u) >> 8;
    if (rnd) a = ((h % (60000 / (rnd))) * rnd) + mis;
    uint32_t acc = 0;
    uint8_t* g = gbuf + (size_t)blockIdx.x * 1048576;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t b = a;
            if (MODE == RD32) acc ^= LDS(uint32_t, b + u * 256 * (rnd ? 0 : 1));
            if (MODE == RD64) { u32x2 v = LDS(u32x2, b + u * 512 * (rnd ? 0 : 1)); acc ^= v.x ^ v.y; }
            if (MODE == RD128) { u32x4 v = LDS(u32x4, b + u * 1024 * (rnd ? 0 : 1) % 4096); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
            if (MODE == RDU8) acc ^= LDS(uint8_t, b + u * 64);
            if (MODE == WR8) LDS(uint8_t, b + u * 64) = (uint8_t)(acc + u);
            if (MODE == WR32) LDS(uint32_t, b + u * 256 * (rnd ? 0 : 1)) = acc + u;
            if (MODE == WR64) { u32x2 v; v.x = acc; v.y = u; LDS(u32x2, b + u * 512 * (rnd ? 0 : 1)) = v; }
            if (MODE == WR128) { u32x4 v; v.x = acc; v.y = u; v.z = it; v.w = lane; LDS(u32x4, b + (u & 3) * 1024 * (rnd ? 0 : 1)) = v; }
            if (MODE == BPERM) acc ^= (uint32_t)__builtin_amdgcn_ds_bpermute((int)((h + u * 4) & 0xFC), (int)(acc + u));
            if (MODE == ATOMMAX) __hip_atomic_fetch_max((uint32_t*)__builtin_assume_aligned(smem + ((b + u * 4 * (rnd ? 1 : 64)) & 0xFFFC), 4), acc + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == OR64) __hip_atomic_fetch_or((unsigned long long*)__builtin_assume_aligned(smem + ((b + u * 8) & 0xFFF8), 8), (unsigned long long)(acc + it) << 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == RD32X2) { acc ^= LDS(uint32_t, b + u * 256 * (rnd ? 0 : 1)); acc ^= LDS(uint32_t, b + 4 + u * 256 * (rnd ? 0 : 1)); }
            if (MODE == GST16) { u32x4 v; v.x = acc; v.y = u; v.z = it; v.w = lane; *(u32x4*)(g + (((size_t)(it * 8 + u) * 16384 + w * 1024 + lane * 16 + mis) & 1048575)) = v; }
            if (MODE == GST1) g[((size_t)(it * 8 + u) * 1024 + w * 64 + lane + mis) & 1048575] = (uint8_t)acc;
            if (MODE == GST4) *(uint32_t*)(g + (((size_t)(it * 8 + u) * 4096 + w * 256 + lane * 4 + mis) & 1048575)) = acc;
        }
        if (rnd) { h = h * 1664525u + 1013904223u; a = (((h >> 8) % (60000 / rnd)) * rnd) + mis; }
    }
    out[blockIdx.x * 1024 + tid] = acc;
}

typedef void (*kern_t)(uint32_t, uint32_t, uint32_t, int, uint32_t*, uint8_t*);
template <int M> static kern_t get() { return k<M>; }
static kern_t table[NMODES] = { get<0>(), get<1>(), get<2>(), get<3>(), get<4>(), get<5>(), get<6>(), get<7>(), get<8>(), get<9>(), get<10>(), get<11>(), get<12>(), get<13>(), get<14>() };

int main() {
    uint32_t* d_out; uint8_t* d_g;
    CHK(hipMalloc(&d_out, 256 * 1024 * 4)); CHK(hipMalloc(&d_g, ((size_t)256 << 20) + 4096));      // (+ slack: the misaligned store cases run a few bytes past a block's last megabyte)
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    int clk = 0; CHK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    printf("clock %d kHz\n", clk);
    struct Case { int mode; uint32_t stride, mis, rnd; };
    const Case cases[] = {
        {RD32, 4, 0, 0}, {RD32, 4, 1, 0}, {RD32, 4, 2, 0}, {RD32, 0, 0, 4}, {RD32, 0, 1, 4}, {RD32, 2, 0, 0} /* stride 2: misaligned odd lanes */, {RD32, 8, 0, 0}, {RD32, 8, 3, 0},
        {RD32X2, 0, 0, 4}, {RD32X2, 8, 0, 0},
        {RD64, 8, 0, 0}, {RD64, 8, 1, 0}, {RD64, 8, 4, 0}, {RD64, 0, 0, 8}, {RD64, 0, 3, 8},
        {RD128, 16, 0, 0}, {RD128, 16, 1, 0}, {RD128, 16, 4, 0}, {RD128, 16, 8, 0}, {RD128, 0, 0, 16}, {RD128, 0, 5, 16}, {RD128, 8, 0, 0},
        {RDU8, 1, 0, 0}, {RDU8, 0, 0, 1},
        {WR8, 1, 0, 0}, {WR8, 0, 0, 1}, {WR8, 4, 0, 0},
        {WR32, 4, 0, 0}, {WR32, 4, 1, 0}, {WR32, 0, 0, 4}, {WR32, 0, 1, 4},
        {WR64, 8, 0, 0}, {WR64, 8, 3, 0}, {WR64, 0, 0, 8},
        {WR128, 16, 0, 0}, {WR128, 16, 5, 0}, {WR128, 0, 0, 16}, {WR128, 0, 7, 16},
        {BPERM, 0, 0, 0}, {ATOMMAX, 4, 0, 0}, {ATOMMAX, 0, 0, 4}, {OR64, 8, 0, 0}, {OR64, 0, 0, 8},
        {GST16, 0, 0, 0}, {GST16, 0, 5, 0}, {GST1, 0, 0, 0}, {GST4, 0, 0, 0}, {GST4, 0, 1, 0},
    };
    const int iters = 2000;
    for (const Case& c : cases) {
        kern_t f = table[c.mode];
        CHK(hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, 70000));
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(f, dim3(256), dim3(1024), 70000, 0, c.stride, c.mis, c.rnd, iters, d_out, d_g);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double ops = (double)iters * 8 * 16 * (c.mode == RD32X2 ? 2 : 1);
        printf("%-26s stride %2u mis %u %s: %.3f ms  %.2f cycles per wave-op per CU\n", names[c.mode], c.stride, c.mis, c.rnd ? "random" : "linear",
               best, best * 1e-3 * (double)clk * 1e3 / ops);
    }
    return 0;
}
