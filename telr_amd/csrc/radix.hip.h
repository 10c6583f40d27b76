// Device-wide stable LSD radix sort, hand-written for gfx950 (round 6): the hash-index build's sort of the target minimizers by hash
// (carrying their positions) and of the occurrence counts -- minimap2's `radix_sort_128x` over the index's minimizers inside
// `mm_idx_gen` (the runs behind TELR_alignment.py:69-82, TELR_liftover.py:253-266) -- were rocPRIM's until round 5 (SURVEY 7
// step 5: "write one, keep rocPRIM as a test cross-check": TELR_AB=index_sort_lib keeps the library's, tests/test_gpu_switches.py
// holds the two to the same index).
//
// Eight bits per pass.  A pass is four kernels:
//   k_rs_hist     one workgroup per tile of 2,048 keys: the tile's digit histogram, written digit-major (hist[d * ntiles + tile]) so
//                 that ONE exclusive scan of the flat array gives every (digit, tile) its first output position;
//   k_rs_rowsum / k_rs_scan  that scan, one workgroup per digit (its row's total, then the row in place on top of the smaller digits' totals);
//   k_rs_scatter  the tile again: keys in index order, 256 at a time (a wave holds 64 consecutive keys); a key's rank among the equal
//                 digits of its wave comes from eight ballots (one per digit bit: the lanes that agree with it in every bit), the waves'
//                 counts are chained through LDS in (round, wave) order -- that is the stability --, the tile is laid out by digit in LDS
//                 and leaves in that order: a digit's keys of a tile are one contiguous run of the output, written by consecutive lanes.
// HBM per pass: the keys (+ values) read twice and written once; integer / byte work, no MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RS_BINS 256
#define RS_T 256               /* threads per workgroup */
#define RS_IPT 8               /* keys per thread */
#define RS_TILE (RS_T * RS_IPT)

template <typename K>
__global__ void __launch_bounds__(RS_T) k_rs_hist(const K *__restrict__ keys, int64_t n, int shift, int32_t ntiles, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[RS_BINS];
    const int tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int64_t i = base + j * RS_T + tid;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & (RS_BINS - 1)], 1u);
    }
    __syncthreads();
    hist[(size_t)tid * ntiles + blockIdx.x] = h[tid];
}

// the scan of the digit-major histogram, one workgroup per digit: k_rs_rowsum leaves every digit's total, k_rs_scan turns row d into
// (keys with a smaller digit) + (keys with digit d in the tiles before)
__global__ void __launch_bounds__(256) k_rs_rowsum(const uint32_t *__restrict__ hist, int32_t ntiles, uint32_t *__restrict__ tot)
{
    __shared__ uint32_t s[4];
    const uint32_t *row = hist + (size_t)blockIdx.x * ntiles;
    uint32_t sum = 0;
    for (int i = threadIdx.x; i < ntiles; i += 256) sum += row[i];
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) tot[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ void __launch_bounds__(256) k_rs_scan(uint32_t *__restrict__ hist, int32_t ntiles, const uint32_t *__restrict__ tot)
{
    __shared__ uint32_t s[256];
    const int tid = threadIdx.x, d = blockIdx.x;
    uint32_t *row = hist + (size_t)d * ntiles;
    // keys with a smaller digit: the 256 totals, summed below d
    s[tid] = tid < d ? tot[tid] : 0u;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) { if (tid < o) s[tid] += s[tid + o]; __syncthreads(); }
    const uint32_t base = s[0];
    __syncthreads();
    // thread t owns the contiguous chunk t of the row
    const int per = (ntiles + 255) / 256, lo = tid * per < ntiles ? tid * per : ntiles, hi = lo + per < ntiles ? lo + per : ntiles;
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += row[i];
    s[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const uint32_t x = tid >= o ? s[tid - o] : 0;
        __syncthreads();
        s[tid] += x;
        __syncthreads();
    }
    uint32_t run = base + s[tid] - sum;
    for (int i = lo; i < hi; ++i) { const uint32_t c = row[i]; row[i] = run; run += c; }
}

template <typename K, bool HAS_V>
__global__ void __launch_bounds__(RS_T) k_rs_scatter(const K *__restrict__ keys, const uint32_t *__restrict__ vals, int64_t n, int shift, int32_t ntiles,
                                                      const uint32_t *__restrict__ goff, K *__restrict__ okeys, uint32_t *__restrict__ ovals)
{
    __shared__ uint32_t wcnt[RS_T / 64][RS_BINS];      // this round's count of every digit in every wave
    __shared__ uint32_t run[RS_BINS];                  // keys of the digit in the rounds before; after the last round: the tile's count
    __shared__ uint32_t texcl[RS_BINS];                // first place of the digit in the tile's LDS layout
    __shared__ uint32_t wsum[RS_T / 64];
    __shared__ K lk[RS_TILE];
    __shared__ uint32_t lv[HAS_V ? RS_TILE : 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    const int cnt = n - base < RS_TILE ? (int)(n - base) : RS_TILE;
    run[tid] = 0;
    K k[RS_IPT]; uint32_t v[RS_IPT], rank[RS_IPT];
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int i = j * RS_T + tid;
        k[j] = i < cnt ? keys[base + i] : (K)0;
        if (HAS_V) v[j] = i < cnt ? vals[base + i] : 0u;
    }
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
#pragma unroll
        for (int z = 0; z < RS_T / 64; ++z) wcnt[z][tid] = 0;
        __syncthreads();
        const bool have = j * RS_T + tid < cnt;
        const uint32_t d = (uint32_t)(k[j] >> shift) & (RS_BINS - 1);
        // the lanes of this wave that hold a key with the same digit
        uint64_t same = __ballot(have);
#pragma unroll
        for (int b = 0; b < 8; ++b) { const uint64_t m = __ballot((d >> b) & 1u); same &= ((d >> b) & 1u) ? m : ~m; }
        const uint32_t below = (uint32_t)__builtin_popcountll(same & ((1ULL << lane) - 1ULL));
        if (have && below == 0) wcnt[wv][d] = (uint32_t)__builtin_popcountll(same);
        __syncthreads();
        // digit `tid`: its keys in the earlier rounds and in the earlier waves of this round
        uint32_t before[RS_T / 64]; uint32_t acc = run[tid];
#pragma unroll
        for (int z = 0; z < RS_T / 64; ++z) { before[z] = acc; acc += wcnt[z][tid]; }
        run[tid] = acc;
        __syncthreads();
#pragma unroll
        for (int z = 0; z < RS_T / 64; ++z) wcnt[z][tid] = before[z];
        __syncthreads();
        rank[j] = have ? wcnt[wv][d] + below : 0u;
        __syncthreads();
    }
    // the tile's layout by digit: exclusive scan of its 256 counts
    {
        const uint32_t c = run[tid];
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(inc, o); if (lane >= o) inc += x; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t pre = 0;
#pragma unroll
        for (int z = 0; z < RS_T / 64; ++z) if (z < wv) pre += wsum[z];
        texcl[tid] = pre + inc - c;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        if (j * RS_T + tid < cnt) {
            const uint32_t d = (uint32_t)(k[j] >> shift) & (RS_BINS - 1), p = texcl[d] + rank[j];
            lk[p] = k[j];
            if (HAS_V) lv[p] = v[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int i = j * RS_T + tid;
        if (i < cnt) {
            const K kk = lk[i];
            const uint32_t d = (uint32_t)(kk >> shift) & (RS_BINS - 1);
            const size_t o = (size_t)goff[(size_t)d * ntiles + blockIdx.x] + (uint32_t)(i - (int)texcl[d]);
            okeys[o] = kk;
            if (HAS_V) ovals[o] = lv[i];
        }
    }
}

// Sorts keys[0 .. n) (and vals) by bits [0, nbits) of the key; the result ends in `keys` / `vals` or in `tkeys` / `tvals` (returned:
// 1 = in the temporaries).  hist: 256 * ceil(n / 2048) + 256 counters.  n < 2^32.
template <typename K, bool HAS_V>
static int radix_sort_passes(K *keys, uint32_t *vals, K *tkeys, uint32_t *tvals, int64_t n, int nbits, uint32_t *hist, hipStream_t st)
{
    const int32_t ntiles = (int32_t)((n + RS_TILE - 1) / RS_TILE);
    int where = 0;
    for (int shift = 0; shift < nbits; shift += 8) {
        const K *ik = where ? tkeys : keys; K *ok = where ? keys : tkeys;
        const uint32_t *iv = where ? tvals : vals; uint32_t *ov = where ? vals : tvals;
        hipLaunchKernelGGL((k_rs_hist<K>), dim3(ntiles), dim3(RS_T), 0, st, ik, n, shift, ntiles, hist);
        hipLaunchKernelGGL(k_rs_rowsum, dim3(RS_BINS), dim3(256), 0, st, hist, ntiles, hist + (size_t)RS_BINS * ntiles);
        hipLaunchKernelGGL(k_rs_scan, dim3(RS_BINS), dim3(256), 0, st, hist, ntiles, hist + (size_t)RS_BINS * ntiles);
        hipLaunchKernelGGL((k_rs_scatter<K, HAS_V>), dim3(ntiles), dim3(RS_T), 0, st, ik, iv, n, shift, ntiles, hist, ok, ov);
        where ^= 1;
    }
    return where;
}
