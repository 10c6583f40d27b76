// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD, for the instruction kinds the DP and
// chaining kernels are made of, at 1 / 2 / 4 / 8 resident waves per SIMD.  Cycles come from s_memtime inside the kernel
// (shader clock ticks), not from an assumed frequency; the wall time of the launch gives the clock the part held.
//
//   hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip && ./valu_rate > profiles/rNN_valu_rate.txt
//
// Every kind is a block of 64 INDEPENDENT-enough instructions (8 accumulators, each instruction depends on the one eight
// places back) written in inline assembly so that the compiler cannot fuse, reorder or drop them.  Occupancy is pinned
// with dynamic LDS: a workgroup of 256 threads (one wave on each of the CU's four SIMDs) asks for 1/W of the CU's
// 160 KB, the grid has exactly 256 x W workgroups, so W waves share every SIMD for the whole measurement.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITERS 2048
#define NINSTR 64

#define ROW8(INS) \
    INS(a0) INS(a1) INS(a2) INS(a3) INS(a4) INS(a5) INS(a6) INS(a7)
#define BLOCK64(INS) ROW8(INS) ROW8(INS) ROW8(INS) ROW8(INS) ROW8(INS) ROW8(INS) ROW8(INS) ROW8(INS)

#define I_FMA(r)      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_ADD(r)      asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_MAXI32(r)   asm volatile("v_max_i32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_PKADD(r)    asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_PKSUB(r)    asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_PKMAX(r)    asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_PKASHR(r)   asm volatile("v_pk_ashrrev_i16 %0, 15, %0" : "+v"(r));
#define I_BFI(r)      asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_ALIGNBIT(r) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(r) : "v"(c));
#define I_PERM(r)     asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_DPP(r)      asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r));
#define I_MAX3(r)     asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_MIN3U(r)    asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_SAD(r)      asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_MAD24(r)    asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_CVT(r)      asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r));
#define I_AND(r)      asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_LSHL(r)     asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r));
#define I_PKLSHL(r)   asm volatile("v_pk_lshlrev_b16 %0, 1, %0" : "+v"(r));
#define I_PKMIN(r)    asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_PKMAD(r)    asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_XOR(r)      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_AND_OR(r)   asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_MOV(r)      asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(c));
#define I_READLANE(r) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s0) : "v"(r));
#define I_MAXF32(r)   asm volatile("v_max_f32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_ADDF32(r)   asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_MULF32(r)   asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_SUBU32(r)   asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_OR(r)       asm volatile("v_or_b32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_CNDMASK(r)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(c) : "vcc");
#define I_CMP(r)      asm volatile("v_cmp_gt_i32 vcc, %0, %1" :: "v"(r), "v"(c) : "vcc");
#define I_PKADDF16(r) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_PKMAXF16(r) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_PKFMAF16(r) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_MAXF16(r)   asm volatile("v_max_f16 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_ADD3(r)     asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_LSHLADD(r)  asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r) : "v"(c));
#define I_MAXU32(r)   asm volatile("v_max_u32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_MAXI16(r)   asm volatile("v_max_i16 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_ASHR(r)     asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r));
#define I_LSHR(r)     asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r));
#define I_BFE(r)      asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(r));
#define I_OR3(r)      asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_MED3(r)     asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_SUBREV(r)   asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_ADDU16(r)   asm volatile("v_add_u16 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_XNOR(r)     asm volatile("v_xnor_b32 %0, %0, %1" : "+v"(r) : "v"(c));
#define I_NOT(r)      asm volatile("v_not_b32 %0, %0" : "+v"(r));
#define I_ADDCO(r)    asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r) : "v"(c) : "vcc");
#define I_FMAMIX(r)   asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
#define I_DOT2(r)     asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(r) : "v"(c), "v"(d));
#define I_PKADDF32(r) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##r) : "v"(pc));
#define I_PKMOV(r)    asm volatile("v_pk_mov_b32 %0, %1, %1" : "+v"(p##r) : "v"(pc));

enum { K_FMA, K_ADD, K_MAXI32, K_PKADD, K_PKSUB, K_PKMAX, K_PKASHR, K_BFI, K_ALIGNBIT, K_PERM, K_DPP, K_MAX3, K_MIN3U, K_SAD, K_MAD24,
       K_CVT, K_AND, K_LSHL, K_PKLSHL, K_PKMIN, K_PKMAD, K_XOR, K_AND_OR, K_MOV, K_READLANE,
       K_MAXF32, K_ADDF32, K_MULF32, K_SUBU32, K_OR, K_CNDMASK, K_CMP, K_PKADDF16, K_PKMAXF16, K_PKFMAF16, K_MAXF16, K_ADD3, K_LSHLADD, K_MAXU32, K_MAXI16,
       K_ASHR, K_LSHR, K_BFE, K_OR3, K_MED3, K_SUBREV, K_ADDU16, K_XNOR, K_NOT, K_ADDCO, K_FMAMIX, K_DOT2, K_PKADDF32, K_PKMOV, K_N };
static const char *NAMES[K_N] = { "v_fma_f32 (control)", "v_add_u32", "v_max_i32", "v_pk_add_i16", "v_pk_sub_i16", "v_pk_max_i16", "v_pk_ashrrev_i16",
    "v_bfi_b32", "v_alignbit_b32", "v_perm_b32", "v_mov_b32_dpp wave_shr:1", "v_max3_i32", "v_min3_u32", "v_sad_u32", "v_mad_u32_u24",
    "v_cvt_f32_u32", "v_and_b32", "v_lshlrev_b32", "v_pk_lshlrev_b16", "v_pk_min_i16", "v_pk_mad_i16", "v_xor_b32", "v_and_or_b32", "v_mov_b32", "v_readlane_b32",
    "v_max_f32", "v_add_f32", "v_mul_f32", "v_sub_u32", "v_or_b32", "v_cndmask_b32 (vcc)", "v_cmp_gt_i32 -> vcc", "v_pk_add_f16", "v_pk_max_f16", "v_pk_fma_f16", "v_max_f16", "v_add3_u32", "v_lshl_add_u32",
    "v_max_u32", "v_max_i16", "v_ashrrev_i32", "v_lshrrev_b32", "v_bfe_u32", "v_or3_b32", "v_med3_i32", "v_subrev_u32", "v_add_u16", "v_xnor_b32", "v_not_b32", "v_add_co_u32", "v_fma_mix_f32", "v_dot2_i32_i16",
    "v_pk_add_f32 (2 regs)", "v_pk_mov_b32 (2 regs)" };

template <int KIND> __global__ void __launch_bounds__(256) k(uint64_t *cycles, uint32_t *sink, uint32_t seed)
{
    extern __shared__ uint32_t lds_pad[];
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t c = seed | 1, d = seed * 7 + 3;
    uint32_t s0 = 0;
    uint64_t pa0 = a0, pa1 = a1, pa2 = a2, pa3 = a3, pa4 = a4, pa5 = a5, pa6 = a6, pa7 = a7, pc = ((uint64_t)c << 32) | d;
    __builtin_amdgcn_s_barrier();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if (KIND == K_FMA) { BLOCK64(I_FMA) } else if (KIND == K_ADD) { BLOCK64(I_ADD) } else if (KIND == K_MAXI32) { BLOCK64(I_MAXI32) }
        else if (KIND == K_PKADD) { BLOCK64(I_PKADD) } else if (KIND == K_PKSUB) { BLOCK64(I_PKSUB) } else if (KIND == K_PKMAX) { BLOCK64(I_PKMAX) }
        else if (KIND == K_PKASHR) { BLOCK64(I_PKASHR) } else if (KIND == K_BFI) { BLOCK64(I_BFI) } else if (KIND == K_ALIGNBIT) { BLOCK64(I_ALIGNBIT) }
        else if (KIND == K_PERM) { BLOCK64(I_PERM) } else if (KIND == K_DPP) { BLOCK64(I_DPP) } else if (KIND == K_MAX3) { BLOCK64(I_MAX3) }
        else if (KIND == K_MIN3U) { BLOCK64(I_MIN3U) } else if (KIND == K_SAD) { BLOCK64(I_SAD) } else if (KIND == K_MAD24) { BLOCK64(I_MAD24) }
        else if (KIND == K_CVT) { BLOCK64(I_CVT) } else if (KIND == K_AND) { BLOCK64(I_AND) } else if (KIND == K_LSHL) { BLOCK64(I_LSHL) }
        else if (KIND == K_PKLSHL) { BLOCK64(I_PKLSHL) } else if (KIND == K_PKMIN) { BLOCK64(I_PKMIN) } else if (KIND == K_PKMAD) { BLOCK64(I_PKMAD) }
        else if (KIND == K_XOR) { BLOCK64(I_XOR) } else if (KIND == K_AND_OR) { BLOCK64(I_AND_OR) } else if (KIND == K_MOV) { BLOCK64(I_MOV) }
        else if (KIND == K_READLANE) { BLOCK64(I_READLANE) }
        else if (KIND == K_MAXF32) { BLOCK64(I_MAXF32) } else if (KIND == K_ADDF32) { BLOCK64(I_ADDF32) } else if (KIND == K_MULF32) { BLOCK64(I_MULF32) }
        else if (KIND == K_SUBU32) { BLOCK64(I_SUBU32) } else if (KIND == K_OR) { BLOCK64(I_OR) } else if (KIND == K_CNDMASK) { BLOCK64(I_CNDMASK) }
        else if (KIND == K_CMP) { BLOCK64(I_CMP) } else if (KIND == K_PKADDF16) { BLOCK64(I_PKADDF16) } else if (KIND == K_PKMAXF16) { BLOCK64(I_PKMAXF16) }
        else if (KIND == K_PKFMAF16) { BLOCK64(I_PKFMAF16) } else if (KIND == K_MAXF16) { BLOCK64(I_MAXF16) } else if (KIND == K_ADD3) { BLOCK64(I_ADD3) }
        else if (KIND == K_LSHLADD) { BLOCK64(I_LSHLADD) } else if (KIND == K_MAXU32) { BLOCK64(I_MAXU32) } else if (KIND == K_MAXI16) { BLOCK64(I_MAXI16) }
        else if (KIND == K_ASHR) { BLOCK64(I_ASHR) } else if (KIND == K_LSHR) { BLOCK64(I_LSHR) } else if (KIND == K_BFE) { BLOCK64(I_BFE) }
        else if (KIND == K_OR3) { BLOCK64(I_OR3) } else if (KIND == K_MED3) { BLOCK64(I_MED3) } else if (KIND == K_SUBREV) { BLOCK64(I_SUBREV) }
        else if (KIND == K_ADDU16) { BLOCK64(I_ADDU16) } else if (KIND == K_XNOR) { BLOCK64(I_XNOR) } else if (KIND == K_NOT) { BLOCK64(I_NOT) }
        else if (KIND == K_ADDCO) { BLOCK64(I_ADDCO) } else if (KIND == K_FMAMIX) { BLOCK64(I_FMAMIX) } else if (KIND == K_DOT2) { BLOCK64(I_DOT2) }
        else if (KIND == K_PKADDF32) { BLOCK64(I_PKADDF32) } else if (KIND == K_PKMOV) { BLOCK64(I_PKMOV) }
    }
    asm volatile("s_nop 0" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) cycles[wave] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ lds_pad[0] ^ (uint32_t)(pa0 ^ pa1 ^ pa2 ^ pa3 ^ pa4 ^ pa5 ^ pa6 ^ pa7);
}

template <int KIND> static void run(int W, int n_cu, uint64_t *d_cyc, uint32_t *d_sink, std::vector<uint64_t> &h)
{
    const int blocks = n_cu * W;
    const size_t lds = (size_t)(160 * 1024 / W) - 1024;      // at most W workgroups fit a CU
    hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), lds, 0, d_cyc, d_sink, 12345u);      // warm-up (clock ramp)
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), lds, 0, d_cyc, d_sink, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const int nw = blocks * 4;
    hipMemcpy(h.data(), d_cyc, (size_t)nw * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.begin() + nw);
    const double per = (double)ITERS * NINSTR;
    const double med = (double)h[nw / 2] / per, mn = (double)h[0] / per, mx = (double)h[nw - 1] / per;
    // The waves of a SIMD do not progress evenly (the arbiter favours the oldest wave), so the SIMD's issue interval is
    // the time of its SLOWEST wave over the W x instructions it issued in that time; cross-check: the launch's wall time
    // at the clock implied by the slowest wave's tick count.
    const double ghz = (double)h[nw - 1] / (ms * 1e-3) / 1e9;
    printf("%-26s W=%d  cycles per instruction of one wave: median %6.2f  min %6.2f  max %6.2f   => %5.2f cycles per wave-instruction per SIMD   launch %.3f ms, clock >= %.2f GHz\n",
           NAMES[KIND], W, med, mn, mx, mx / W, ms, ghz);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int KIND> static void run_all(int n_cu, uint64_t *d_cyc, uint32_t *d_sink, std::vector<uint64_t> &h)
{
    for (int W : { 1, 2, 4, 8 }) run<KIND>(W, n_cu, d_cyc, d_sink, h);
}

int main()
{
    hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int n_cu = p.multiProcessorCount;
    printf("# %s (%s), %d CUs; %d iterations x %d instructions per wave; s_memtime ticks = shader cycles\n", p.name, p.gcnArchName, n_cu, ITERS, NINSTR);
    printf("# columns: instruction, resident waves per SIMD (W), cycles one wave needs per instruction (median / fastest / slowest wave), the SIMD's issue interval (= slowest / W)\n");
    uint64_t *d_cyc; uint32_t *d_sink;
    hipMalloc(&d_cyc, (size_t)n_cu * 8 * 4 * 8); hipMalloc(&d_sink, (size_t)n_cu * 8 * 256 * 4);
    std::vector<uint64_t> h((size_t)n_cu * 8 * 4);
    run_all<K_FMA>(n_cu, d_cyc, d_sink, h); run_all<K_ADD>(n_cu, d_cyc, d_sink, h); run_all<K_MAXI32>(n_cu, d_cyc, d_sink, h);
    run_all<K_PKADD>(n_cu, d_cyc, d_sink, h); run_all<K_PKSUB>(n_cu, d_cyc, d_sink, h); run_all<K_PKMAX>(n_cu, d_cyc, d_sink, h);
    run_all<K_PKMIN>(n_cu, d_cyc, d_sink, h); run_all<K_PKASHR>(n_cu, d_cyc, d_sink, h); run_all<K_PKLSHL>(n_cu, d_cyc, d_sink, h);
    run_all<K_PKMAD>(n_cu, d_cyc, d_sink, h);
    run_all<K_BFI>(n_cu, d_cyc, d_sink, h); run_all<K_ALIGNBIT>(n_cu, d_cyc, d_sink, h); run_all<K_PERM>(n_cu, d_cyc, d_sink, h);
    run_all<K_DPP>(n_cu, d_cyc, d_sink, h); run_all<K_MAX3>(n_cu, d_cyc, d_sink, h); run_all<K_MIN3U>(n_cu, d_cyc, d_sink, h);
    run_all<K_SAD>(n_cu, d_cyc, d_sink, h); run_all<K_MAD24>(n_cu, d_cyc, d_sink, h); run_all<K_CVT>(n_cu, d_cyc, d_sink, h);
    run_all<K_AND>(n_cu, d_cyc, d_sink, h); run_all<K_XOR>(n_cu, d_cyc, d_sink, h); run_all<K_AND_OR>(n_cu, d_cyc, d_sink, h);
    run_all<K_LSHL>(n_cu, d_cyc, d_sink, h); run_all<K_MOV>(n_cu, d_cyc, d_sink, h); run_all<K_READLANE>(n_cu, d_cyc, d_sink, h);
    run_all<K_MAXF32>(n_cu, d_cyc, d_sink, h); run_all<K_ADDF32>(n_cu, d_cyc, d_sink, h); run_all<K_MULF32>(n_cu, d_cyc, d_sink, h);
    run_all<K_SUBU32>(n_cu, d_cyc, d_sink, h); run_all<K_SUBREV>(n_cu, d_cyc, d_sink, h); run_all<K_OR>(n_cu, d_cyc, d_sink, h); run_all<K_XNOR>(n_cu, d_cyc, d_sink, h); run_all<K_NOT>(n_cu, d_cyc, d_sink, h);
    run_all<K_CNDMASK>(n_cu, d_cyc, d_sink, h); run_all<K_CMP>(n_cu, d_cyc, d_sink, h); run_all<K_ADDCO>(n_cu, d_cyc, d_sink, h);
    run_all<K_PKADDF16>(n_cu, d_cyc, d_sink, h); run_all<K_PKMAXF16>(n_cu, d_cyc, d_sink, h); run_all<K_PKFMAF16>(n_cu, d_cyc, d_sink, h); run_all<K_MAXF16>(n_cu, d_cyc, d_sink, h);
    run_all<K_ADDU16>(n_cu, d_cyc, d_sink, h); run_all<K_MAXI16>(n_cu, d_cyc, d_sink, h);
    run_all<K_ADD3>(n_cu, d_cyc, d_sink, h); run_all<K_LSHLADD>(n_cu, d_cyc, d_sink, h); run_all<K_MAXU32>(n_cu, d_cyc, d_sink, h);
    run_all<K_ASHR>(n_cu, d_cyc, d_sink, h); run_all<K_LSHR>(n_cu, d_cyc, d_sink, h); run_all<K_BFE>(n_cu, d_cyc, d_sink, h); run_all<K_OR3>(n_cu, d_cyc, d_sink, h); run_all<K_MED3>(n_cu, d_cyc, d_sink, h);
    run_all<K_FMAMIX>(n_cu, d_cyc, d_sink, h); run_all<K_DOT2>(n_cu, d_cyc, d_sink, h); run_all<K_PKADDF32>(n_cu, d_cyc, d_sink, h); run_all<K_PKMOV>(n_cu, d_cyc, d_sink, h);
    hipFree(d_cyc); hipFree(d_sink);
    return 0;
}
