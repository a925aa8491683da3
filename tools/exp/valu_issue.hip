// micro-benchmark (throw-away measurement program, not part of the product): how often does one SIMD of gfx950 take a
// wave64 VALU instruction?  1, 2, 4, 8 waves per SIMD, each running chains of one opcode: ILP 1 (every instruction
// depends on the one before: latency) and ILP 8 (eight independent chains per lane: issue rate).
// Prints cycles per wave-instruction per SIMD = the wave's own cycles (s_memtime) / (instructions per wave * waves on the SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue tools/exp/valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

enum { OP_ADD = 0, OP_AND, OP_ALIGNBYTE, OP_MUL_LO, OP_MAD24, OP_DPP_SHR, OP_CNDMASK, OP_LSHL_ADD, OP_SALU_ADD, OP_MIX_VS, OP_DPP_ADD, OP_FFBL, OP_CMP_ADDC, OP_XOR, OP_LSHR, OP_MINU, OP_BFE, OP_PERM, OP_ADD3, OP_ANDOR, OP_ADD_E64, OP_ADD_LIT, OP_MUL24, OP_CMP, OP_CMP_SG, OP_READLANE, OP_MBCNT, OP_MOV, OP_LSHL64, OP_ROWSHR_MAX, OP_SNOP, OP_VS_2_1, OP_V24, OP_CND_E64, OP_CMP_CND, OP_FFSS, OP_OR, OP_SUB, OP_LSHL, OP_MAXU, OP_ADDCO, OP_DSREAD, OP_DS_V, NOPS };
static const char* names[] = { "v_add_u32", "v_and_b32", "v_alignbyte_b32", "v_mul_lo_u32", "v_mad_u32_u24", "v_mov_b32 dpp wave_shr:1", "v_cndmask_b32", "v_lshl_add_u32", "s_add_u32 (scalar)", "v_add_u32 + s_add_u32 interleaved", "v_add_u32 dpp row_shr:1", "v_ffbl_b32", "v_cmp + v_addc (pairs)", "v_xor_b32", "v_lshrrev_b32", "v_min_u32", "v_bfe_u32", "v_perm_b32", "v_add3_u32", "v_and_or_b32", "v_add_u32 (VOP3 encoding, sgpr operand)", "v_add_u32 (32-bit literal)", "v_mul_u32_u24", "v_cmp_lt_u32 (writes vcc)", "v_cmp_lt_u32 (writes sgpr pair)", "v_readlane_b32", "v_mbcnt_lo_u32_b32", "v_mov_b32", "v_lshlrev_b64", "v_max_u32 dpp row_shr:1", "s_nop 0", "2 x v_add_u32 + 1 x s_add_u32", "v_add_u32 + v_alignbyte_b32 (a 2-cycle and a 4-cycle op)", "v_cndmask_b32 (mask in s[20:21], set once by s_mov)", "v_cmp_lt_u32 vcc + v_cndmask_b32 vcc (pairs)", "4 x v_add_u32 then 4 x v_alignbyte_b32 (clustered)", "v_or_b32", "v_sub_u32", "v_lshlrev_b32", "v_max_u32", "v_add_co_u32 (writes vcc)", "ds_read_b32 (same address per lane, no wait)", "ds_read_b32 + 4 x v_add_u32" };

// 64 instructions of one opcode in ONE asm statement (between separate asm statements the compiler puts an s_nop: it cannot see
// what they do): ILP 1: all on register a0; ILP 8: round robin over a0..a7
#define I1(T) T(0) T(0) T(0) T(0) T(0) T(0) T(0) T(0)
#define I8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define R8(X) X X X X X X X X
#define T_ADD(i) "v_add_u32 %" #i ", %" #i ", %9\n\t"
#define T_AND(i) "v_and_b32 %" #i ", %" #i ", %9\n\t"
#define T_ALIGN(i) "v_alignbyte_b32 %" #i ", %" #i ", %9, 1\n\t"
#define T_MUL(i) "v_mul_lo_u32 %" #i ", %" #i ", %9\n\t"
#define T_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %9, %9\n\t"
#define T_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define T_DPPADD(i) "v_add_u32_dpp %" #i ", %" #i ", %9 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define T_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t"
#define T_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %9\n\t"
#define T_SADD(i) "s_add_u32 %8, %8, 3\n\t"
#define T_MIX(i) "v_add_u32 %" #i ", %" #i ", %9\n\ts_add_u32 %8, %8, 3\n\t"
#define T_FFBL(i) "v_ffbl_b32 %" #i ", %" #i "\n\t"
#define T_CMPX(i) "v_cmp_lt_u32 vcc, %" #i ", %9\n\tv_addc_co_u32 %" #i ", vcc, %" #i ", %9, vcc\n\t"
#define T_OP_XOR(i) "v_xor_b32 %" #i ", %" #i ", %9\n\t"
#define T_OP_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n\t"
#define T_OP_MINU(i) "v_min_u32 %" #i ", %" #i ", %9\n\t"
#define T_OP_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 8\n\t"
#define T_OP_PERM(i) "v_perm_b32 %" #i ", %" #i ", %9, %9\n\t"
#define T_OP_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %9, %9\n\t"
#define T_OP_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %9, %9\n\t"
#define T_OP_ADD_E64(i) "v_add_u32_e64 %" #i ", %" #i ", %8\n\t"
#define T_OP_ADD_LIT(i) "v_add_u32 %" #i ", 0x12345, %" #i "\n\t"
#define T_OP_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %9\n\t"
#define T_OP_CMP(i) "v_cmp_lt_u32 vcc, %" #i ", %9\n\t"
#define T_OP_CMP_SG(i) "v_cmp_lt_u32_e64 s[20:21], %" #i ", %9\n\t"
#define T_OP_READLANE(i) "v_readlane_b32 s20, %" #i ", 5\n\t"
#define T_OP_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %9, %" #i "\n\t"
#define T_OP_MOV(i) "v_mov_b32 %" #i ", %9\n\t"
#define T_OP_LSHL64(i) "v_lshlrev_b64 v[40:41], 3, v[42:43]\n\t"
#define T_OP_ROWSHR_MAX(i) "v_max_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define T_OP_SNOP(i) "s_nop 0\n\t"
#define T_OP_VS_2_1(i) "v_add_u32 %" #i ", %" #i ", %9\n\tv_xor_b32 %" #i ", %" #i ", %9\n\ts_add_u32 %8, %8, 3\n\t"
#define T_OP_V24(i) "v_add_u32 %" #i ", %" #i ", %9\n\tv_alignbyte_b32 %" #i ", %" #i ", %9, 1\n\t"
#define T_OP_CND_E64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[20:21]\n\t"
#define T_OP_CMP_CND(i) "v_cmp_lt_u32 vcc, %" #i ", %9\n\tv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t"
#define T_OP_FFSS(i) "v_add_u32 %" #i ", %" #i ", %9\n\t"
#define T_OP_OR(i) "v_or_b32 %" #i ", %" #i ", %9\n\t"
#define T_OP_SUB(i) "v_sub_u32 %" #i ", %" #i ", %9\n\t"
#define T_OP_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n\t"
#define T_OP_MAXU(i) "v_max_u32 %" #i ", %" #i ", %9\n\t"
#define T_OP_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %9\n\t"
#define T_OP_DSREAD(i) "ds_read_b32 %" #i ", %9\n\t"
#define T_OP_DS_V(i) "ds_read_b32 v40, %9\n\tv_add_u32 %" #i ", %" #i ", %9\n\tv_xor_b32 %" #i ", %" #i ", %9\n\tv_add_u32 %" #i ", %" #i ", %9\n\tv_xor_b32 %" #i ", %" #i ", %9\n\t"
#define ASM64(BODY) asm volatile(BODY : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) , "+s"(s) : "v"(x) : "vcc", "scc", "s20", "s21", "v40", "v41", "v42", "v43")
template <int OP, int ILP> __device__ __forceinline__ void block64(uint32_t (&a)[8], uint32_t x, uint32_t& s) {
#define BOTH(T) do { if (ILP == 1) ASM64(R8(I1(T))); else ASM64(R8(I8(T))); } while (0)
    if (OP == OP_ADD) BOTH(T_ADD);
    if (OP == OP_AND) BOTH(T_AND);
    if (OP == OP_ALIGNBYTE) BOTH(T_ALIGN);
    if (OP == OP_MUL_LO) BOTH(T_MUL);
    if (OP == OP_MAD24) BOTH(T_MAD24);
    if (OP == OP_DPP_SHR) BOTH(T_DPP);
    if (OP == OP_CNDMASK) BOTH(T_CND);
    if (OP == OP_LSHL_ADD) BOTH(T_LSHLADD);
    if (OP == OP_SALU_ADD) BOTH(T_SADD);
    if (OP == OP_MIX_VS) BOTH(T_MIX);
    if (OP == OP_DPP_ADD) BOTH(T_DPPADD);
    if (OP == OP_FFBL) BOTH(T_FFBL);
    if (OP == OP_CMP_ADDC) BOTH(T_CMPX);
    if (OP == OP_XOR) BOTH(T_OP_XOR);
    if (OP == OP_LSHR) BOTH(T_OP_LSHR);
    if (OP == OP_MINU) BOTH(T_OP_MINU);
    if (OP == OP_BFE) BOTH(T_OP_BFE);
    if (OP == OP_PERM) BOTH(T_OP_PERM);
    if (OP == OP_ADD3) BOTH(T_OP_ADD3);
    if (OP == OP_ANDOR) BOTH(T_OP_ANDOR);
    if (OP == OP_ADD_E64) BOTH(T_OP_ADD_E64);
    if (OP == OP_ADD_LIT) BOTH(T_OP_ADD_LIT);
    if (OP == OP_MUL24) BOTH(T_OP_MUL24);
    if (OP == OP_CMP) BOTH(T_OP_CMP);
    if (OP == OP_CMP_SG) BOTH(T_OP_CMP_SG);
    if (OP == OP_READLANE) BOTH(T_OP_READLANE);
    if (OP == OP_MBCNT) BOTH(T_OP_MBCNT);
    if (OP == OP_MOV) BOTH(T_OP_MOV);
    if (OP == OP_LSHL64) BOTH(T_OP_LSHL64);
    if (OP == OP_ROWSHR_MAX) BOTH(T_OP_ROWSHR_MAX);
    if (OP == OP_SNOP) BOTH(T_OP_SNOP);
    if (OP == OP_VS_2_1) BOTH(T_OP_VS_2_1);
    if (OP == OP_V24) BOTH(T_OP_V24);
    if (OP == OP_CND_E64) BOTH(T_OP_CND_E64);
    if (OP == OP_CMP_CND) BOTH(T_OP_CMP_CND);
    if (OP == OP_OR) BOTH(T_OP_OR);
    if (OP == OP_SUB) BOTH(T_OP_SUB);
    if (OP == OP_LSHL) BOTH(T_OP_LSHL);
    if (OP == OP_MAXU) BOTH(T_OP_MAXU);
    if (OP == OP_ADDCO) BOTH(T_OP_ADDCO);
    if (OP == OP_DSREAD) BOTH(T_OP_DSREAD);
    if (OP == OP_DS_V) BOTH(T_OP_DS_V);
    if (OP == OP_FFSS) ASM64(R8("v_add_u32 %0, %0, %9\n\tv_add_u32 %1, %1, %9\n\tv_add_u32 %2, %2, %9\n\tv_add_u32 %3, %3, %9\n\tv_alignbyte_b32 %4, %4, %9, 1\n\tv_alignbyte_b32 %5, %5, %9, 1\n\tv_alignbyte_b32 %6, %6, %9, 1\n\tv_alignbyte_b32 %7, %7, %9, 1\n\t"));
}

template <int OP, int ILP>
__global__ void __launch_bounds__(1024) k(int iters, uint32_t* out, unsigned long long* cyc) {
    uint32_t a[8], s = blockIdx.x;
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i;
    const uint32_t x = threadIdx.x * 4u + 4u;      // (an aligned LDS address for the ds_read cases, any value for the others)
    asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) block64<OP, ILP>(a, x, s);
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = s;
    for (int i = 0; i < 8; i++) acc ^= a[i];
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(int, uint32_t*, unsigned long long*);
template <int OP> static void run_op(uint32_t* d_out, unsigned long long* d_cyc, int ncu) {
    const int iters = 2000;
    const double n_inst = (double)iters * 64 * ((OP == OP_MIX_VS || OP == OP_CMP_ADDC || OP == OP_V24) ? 2 : OP == OP_VS_2_1 ? 3 : OP == OP_CMP_CND ? 2 : OP == OP_DS_V ? 5 : 1);
    for (int ilp = 0; ilp < 2; ilp++) {
        kern_t fn = ilp == 0 ? (kern_t)k<OP, 1> : (kern_t)k<OP, 8>;
        for (int wps = 1; wps <= 8; wps *= 2) {
            if (ilp == 0 && wps != 1 && wps != 4) continue;      // (ILP 1: latency alone and four waves)
            // wps waves per SIMD: one workgroup of 256 * wps threads per CU (two of 1024 for 8)
            const int threads = wps <= 4 ? 256 * wps : 1024, wgs_per_cu = wps <= 4 ? 1 : 2;
            const int grid = ncu * wgs_per_cu;
            hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
            hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), 8192, 0, 10, d_out, d_cyc);       // warm
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), 8192, 0, iters, d_out, d_cyc);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            static unsigned long long h[512 * 16];
            CHK(hipMemcpy(h, d_cyc, sizeof(unsigned long long) * grid * 16, hipMemcpyDeviceToHost));
            double sum = 0, mx = 0; int nw = 0;
            for (int b = 0; b < grid; b++) for (int w = 0; w < threads / 64; w++) { const double c = (double)h[b * 16 + w]; sum += c; if (c > mx) mx = c; nw++; }
            const double mean = sum / nw;
            // readcyclecounter (s_memtime) ticks at a constant 100 MHz on this part: convert with the kernel's wall time
            printf("%-34s ILP %d  %d wave(s)/SIMD: %9.3f ms  per-wave ticks mean %.0f max %.0f  -> %.2f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz (wall)\n",
                   names[OP], ilp == 0 ? 1 : 8, wps, ms, mean, mx, ms * 1e6 / (n_inst * wps), ms * 1e6 / (n_inst * wps) * 2.4);
        }
    }
}

int main(int argc, char**) {
    int dev = 0; CHK(hipSetDevice(dev));
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, dev));
    printf("%s: %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    uint32_t* d_out; unsigned long long* d_cyc;
    CHK(hipMalloc(&d_out, 512 * 1024 * 4)); CHK(hipMalloc(&d_cyc, 512 * 16 * 8));
    const int ncu = p.multiProcessorCount;
    if (argc > 1) {
    run_op<OP_ADD>(d_out, d_cyc, ncu);
    run_op<OP_AND>(d_out, d_cyc, ncu);
    run_op<OP_ALIGNBYTE>(d_out, d_cyc, ncu);
    run_op<OP_MUL_LO>(d_out, d_cyc, ncu);
    run_op<OP_MAD24>(d_out, d_cyc, ncu);
    run_op<OP_DPP_SHR>(d_out, d_cyc, ncu);
    run_op<OP_CNDMASK>(d_out, d_cyc, ncu);
    run_op<OP_LSHL_ADD>(d_out, d_cyc, ncu);
    run_op<OP_SALU_ADD>(d_out, d_cyc, ncu);
    run_op<OP_MIX_VS>(d_out, d_cyc, ncu);
    run_op<OP_DPP_ADD>(d_out, d_cyc, ncu);
    run_op<OP_FFBL>(d_out, d_cyc, ncu);
    run_op<OP_CMP_ADDC>(d_out, d_cyc, ncu);
    run_op<OP_XOR>(d_out, d_cyc, ncu);
    run_op<OP_LSHR>(d_out, d_cyc, ncu);
    run_op<OP_MINU>(d_out, d_cyc, ncu);
    run_op<OP_BFE>(d_out, d_cyc, ncu);
    run_op<OP_PERM>(d_out, d_cyc, ncu);
    run_op<OP_ADD3>(d_out, d_cyc, ncu);
    run_op<OP_ANDOR>(d_out, d_cyc, ncu);
    run_op<OP_ADD_E64>(d_out, d_cyc, ncu);
    run_op<OP_ADD_LIT>(d_out, d_cyc, ncu);
    run_op<OP_MUL24>(d_out, d_cyc, ncu);
    run_op<OP_CMP>(d_out, d_cyc, ncu);
    run_op<OP_CMP_SG>(d_out, d_cyc, ncu);
    run_op<OP_READLANE>(d_out, d_cyc, ncu);
    run_op<OP_MBCNT>(d_out, d_cyc, ncu);
    run_op<OP_MOV>(d_out, d_cyc, ncu);
    run_op<OP_LSHL64>(d_out, d_cyc, ncu);
    run_op<OP_ROWSHR_MAX>(d_out, d_cyc, ncu);
    run_op<OP_SNOP>(d_out, d_cyc, ncu);
    run_op<OP_VS_2_1>(d_out, d_cyc, ncu);
    run_op<OP_V24>(d_out, d_cyc, ncu);
    }
    run_op<OP_CND_E64>(d_out, d_cyc, ncu);
    run_op<OP_CMP_CND>(d_out, d_cyc, ncu);
    run_op<OP_FFSS>(d_out, d_cyc, ncu);
    run_op<OP_OR>(d_out, d_cyc, ncu);
    run_op<OP_SUB>(d_out, d_cyc, ncu);
    run_op<OP_LSHL>(d_out, d_cyc, ncu);
    run_op<OP_MAXU>(d_out, d_cyc, ncu);
    run_op<OP_ADDCO>(d_out, d_cyc, ncu);
    run_op<OP_DSREAD>(d_out, d_cyc, ncu);
    run_op<OP_DS_V>(d_out, d_cyc, ncu);
    return 0;
}
