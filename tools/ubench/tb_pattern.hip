// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the two access patterns of the packed DP path (the guide only
// calibrates 16-B-per-lane streaming accesses):
//   k_read_lines   every lane reads ONE whole 64-byte line of its own (4 x 16 B), lines scattered   -> k_traceback_pk
//   k_write_chunks every lane appends 40-byte pieces (5 x 8 B) to its own contiguous region          -> k_dp_pk, R = 5
// Known byte counts: 1 GiB each.  Build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/tb_pattern tools/ubench/tb_pattern.hip
// Run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./tools/ubench/tb_pattern   (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ void k_read_lines(const uint4 *__restrict__ src, uint32_t n_lines, uint32_t *__restrict__ out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_lines) return;
    const uint32_t line = (uint32_t)(((uint64_t)t * 2654435761u) % n_lines);     // scattered, every line once (n_lines odd-free: power of two, odd multiplier)
    const uint4 *p = src + (size_t)line * 4;
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    const uint32_t s = a.x ^ a.w ^ b.y ^ c.z ^ d.w;
    if (s == 0x12345678u) out[t & 1023] = s;
}

// the same with 128 / 256 contiguous bytes per lane (one or two full 128-byte blocks)
template <int Q>     // uint4 per lane
__global__ void k_read_blocks(const uint4 *__restrict__ src, uint32_t n_blocks, uint32_t *__restrict__ out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_blocks) return;
    const uint32_t blk = (uint32_t)(((uint64_t)t * 2654435761u) % n_blocks);
    const uint4 *p = src + (size_t)blk * Q;
    uint32_t s = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q) { const uint4 v = p[q]; s ^= v.x ^ v.w; }
    if (s == 0x12345678u) out[t & 1023] = s;
}

__global__ void k_write_chunks(uint2 *__restrict__ dst, uint32_t n_threads, int rows)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_threads) return;
    uint2 *p = dst + (size_t)t * rows * 5;
    for (int k = 0; k < rows; ++k) {
#pragma unroll
        for (int r = 0; r < 5; ++r) p[k * 5 + r] = make_uint2(t + k, r);
        // something to do between two pieces, as the DP kernel has
        for (int z = 0; z < 64; ++z) asm volatile("v_mov_b32 %0, %0" : "+v"(k));
    }
}

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 1) << 30;     // GiB read (the write test stays at its share of it)
    void *a, *b; uint32_t *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4096);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipDeviceSynchronize();
    const uint32_t n_lines = (uint32_t)(bytes / 64);
    hipLaunchKernelGGL(k_read_lines, dim3(n_lines / 256), dim3(256), 0, 0, (const uint4*)a, n_lines, o);
    hipLaunchKernelGGL((k_read_blocks<8>), dim3(n_lines / 2 / 256), dim3(256), 0, 0, (const uint4*)a, n_lines / 2, o);
    hipLaunchKernelGGL((k_read_blocks<16>), dim3(n_lines / 4 / 256), dim3(256), 0, 0, (const uint4*)a, n_lines / 4, o);
    const int rows = 64; const uint32_t nt = (uint32_t)(bytes / (40 * rows));
    hipLaunchKernelGGL(k_write_chunks, dim3((nt + 63) / 64), dim3(64), 0, 0, (uint2*)b, nt, rows);
    hipDeviceSynchronize();
    printf("read %zu bytes in 64-byte lines; wrote %zu bytes in 40-byte pieces\n", (size_t)n_lines * 64, (size_t)nt * rows * 40);
    return 0;
}
