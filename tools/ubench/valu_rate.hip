// VALU issue-rate microbenchmark: cycles per wave-instruction per SIMD for the instruction kinds the DP kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s2 __attribute__((ext_vector_type(2)));
#define REP 64
#define ITERS 4096
template <int KIND> __global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 77;
    uint32_t c = seed | 1;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) a[i] = a[i] + c;                                                   // v_add_u32
                else if (KIND == 1) a[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2, a[i]) + __builtin_bit_cast(s2, c));          // v_pk_add_u16
                else if (KIND == 2) a[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s2, a[i]), __builtin_bit_cast(s2, c)));  // v_pk_max_i16
                else if (KIND == 3) a[i] = max((int)a[i], (int)c) + 1;                             // v_max_i32 + add (2 instr)
                else if (KIND == 4) a[i] = __builtin_amdgcn_alignbit(a[i], c, 16);                 // v_alignbit_b32
                else if (KIND == 5) a[i] = __builtin_amdgcn_perm(a[i], c, 0x0c0c0200u) + 1;         // v_perm_b32 + add
                else if (KIND == 6) a[i] = __builtin_amdgcn_update_dpp(0, (int)a[i], 0x138, 0xf, 0xf, false) + 1;   // dpp mov + add
                else if (KIND == 7) a[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2, a[i]) >> (s2)(15)) + c;   // v_pk_ashrrev + add
                else if (KIND == 8) a[i] = (a[i] > c) ? a[i] - c : a[i] + 3;                       // cmp + cndmask + ...
            }
        }
        c += 3;
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char *name, int instr_per_op)
{
    uint32_t *d; hipMalloc(&d, 256 * 8192 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * 8;    // 8 blocks x 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves_per_simd = blocks * 4.0 / (256 * 4);
    double ops = (double)ITERS * REP * waves_per_simd;        // wave-ops per SIMD
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s %8.3f ms   %.2f cycles(@2.4GHz) per wave-op per SIMD  (%d instr/op -> %.2f cyc/instr)\n", name, ms, cyc / ops, instr_per_op, cyc / ops / instr_per_op);
    hipFree(d);
}
int main()
{
    run<0>("v_add_u32", 1); run<1>("v_pk_add_u16", 1); run<2>("v_pk_max_i16", 1); run<3>("v_max_i32+v_add", 2);
    run<4>("v_alignbit_b32", 1); run<5>("v_perm_b32+add", 2); run<6>("v_mov_dpp wave_shr+add", 2); run<7>("v_pk_ashrrev_i16+add", 2); run<8>("cmp+cndmask+sub/add", 4);
    return 0;
}
