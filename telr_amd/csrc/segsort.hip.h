// telr_amd/csrc/segsort.hip.h -- hand-written segmented sort of 64-bit keys, one workgroup per segment, LDS-resident.
//
// Stands for the per-read `radix_sort_128x` of minimap2's seeding inside the aligner runs of TELR_alignment.py:69-82 (and
// for the two smaller per-read orderings of this engine: peaks by score, chains by score).  Replaces rocPRIM's segmented
// radix sort on the hot path (round 3: four radix passes over key + value through HBM plus a join kernel, 42 ms of kernel
// time per configs[2] step): a segment is read ONCE, ordered in LDS on the full 64-bit key and written ONCE.
//
//   k_segsort_classify   one thread per segment: its size class ("tier") -> the tier's segment list (wave-aggregated
//                        atomics); segments above the largest tier go to the fall-back offsets (rocPRIM, rare: a read
//                        with more than 20,480 anchors)
//   k_segsort<T, E>      T threads, E keys per thread, T * E * 8 bytes of LDS.  Coalesced load into LDS, every thread
//                        sorts its E consecutive keys in registers (Batcher's odd-even merge network), then log2(T)
//                        merge rounds: registers -> LDS, merge-path bisection, serial merge of E outputs back into
//                        registers; the rounds stop as soon as one run covers the segment.  Coalesced store.
//                        Persistent over the tier's list (grid-stride), so no host-side counts are needed.
//
// Keys are unique inside a segment (anchor keys) or carry their index in the low bits (peak / chain keys), so stability is
// not an issue; the pad value ~0 never occurs as a key (strand 1, position 2^31 - 1, query position 2^24 - 1, span 255).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SEGSORT_TIERS 7
#define SEGSORT_CAP   20480          /* keys of the largest tier: 1024 threads x 20 keys = all 160 KiB of a CU's LDS */
#define SEGSORT_PAD   0xffffffffffffffffULL

struct SegSortArgs {
    const uint64_t *in; uint64_t *out;
    const int32_t *seg_beg, *seg_end;      // destination range of segment s: out[seg_beg[s] .. seg_end[s])
    const int64_t *src_beg;                // nullable: the segment's keys start at in[src_beg[s]] instead of in[seg_beg[s]]
    const int32_t *order;                  // nullable: visit segments in this order (longest reads first)
    int32_t nseg;
    int32_t *tier_cnt;                     // [SEGSORT_TIERS + 1]: segments per tier, [SEGSORT_TIERS] = fall-back segments
    int32_t *tier_list;                    // [SEGSORT_TIERS][nseg]
    int32_t *fb_beg, *fb_end;              // [nseg] fall-back offsets (empty range unless the segment exceeds SEGSORT_CAP); nullable
};

__host__ __device__ __forceinline__ int segsort_tier_of(int n)
{
    return n <= 128 ? 0 : n <= 512 ? 1 : n <= 1024 ? 2 : n <= 2048 ? 3 : n <= 4096 ? 4 : n <= 8192 ? 5 : n <= SEGSORT_CAP ? 6 : 7;
}

__global__ void __launch_bounds__(256) k_segsort_classify(SegSortArgs A)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int s = -1, tier = -1;
    if (i < A.nseg) {
        s = A.order ? A.order[i] : i;
        const int n = A.seg_end[s] - A.seg_beg[s];
        tier = n <= 0 ? -1 : segsort_tier_of(n);
        if (A.fb_beg) { const bool fb = tier == SEGSORT_TIERS; A.fb_beg[s] = fb ? A.seg_beg[s] : 0; A.fb_end[s] = fb ? A.seg_end[s] : 0; }
    }
    const int lane = threadIdx.x & 63;
    for (int t = 0; t <= SEGSORT_TIERS; ++t) {
        const uint64_t m = __ballot(tier == t);
        if (!m) continue;
        int base = 0;
        if (lane == __builtin_ctzll(m)) base = atomicAdd(&A.tier_cnt[t], __builtin_popcountll(m));
        base = __shfl(base, __builtin_ctzll(m));
        if (tier == t && t < SEGSORT_TIERS) A.tier_list[(size_t)t * A.nseg + base + __builtin_popcountll(m & ((1ULL << lane) - 1))] = s;
    }
}

__device__ __forceinline__ void d_cswap(uint64_t &a, uint64_t &b)
{
    const bool sw = a > b;
    const uint64_t lo = sw ? b : a, hi = sw ? a : b;
    a = lo; b = hi;
}
// Batcher's odd-even merge sort on N registers (any N: the network of the next power of two with the comparators that
// would touch a pad above N left out -- a pad is +inf and never moves)
template <int N> __device__ __forceinline__ void d_regsort(uint64_t (&k)[N])
{
#pragma unroll
    for (int p = 1; p < N; p <<= 1) {
#pragma unroll
        for (int q = p; q >= 1; q >>= 1) {
#pragma unroll
            for (int j = q % p; j + q < N; j += 2 * q) {
#pragma unroll
                for (int i = 0; i < q; ++i) {
                    if (i + j + q < N && (i + j) / (2 * p) == (i + j + q) / (2 * p)) d_cswap(k[i + j], k[i + j + q]);
                }
            }
        }
    }
}

// where a segment's keys come from: copied from an array (coalesced) ...
struct LoadKeys {
    __device__ __forceinline__ void fill(const SegSortArgs &A, int s, uint64_t *lds, int n, int npad, int tid, int nthr) const
    {
        const uint64_t *src = A.in + (A.src_beg ? A.src_beg[s] : (int64_t)A.seg_beg[s]);
        for (int i = tid; i < npad; i += nthr) lds[i] = i < n ? src[i] : SEGSORT_PAD;
    }
};
// ... or made on the spot: the anchor keys of query s written straight into LDS by the seeding routine (kernels.hip.h:
// d_seed_query<1>), so that unsorted keys never exist in HBM (8 B written + 8 B read per anchor less, and the gather latency of
// the occurrence lists hides behind the other workgroups' sorting)
#ifdef TELR_HAVE_SEED_ARGS
struct SeedProducer {
    SeedArgs S;
    __device__ __forceinline__ void fill(const SegSortArgs &A, int s, uint64_t *lds, int n, int npad, int tid, int nthr) const
    {
        SeedArgs B = S; B.lds_keys = lds;
        for (int i = n + tid; i < npad; i += nthr) lds[i] = SEGSORT_PAD;
        d_seed_query<1>(B, s, tid, nthr);
    }
};
#endif

template <int T, int E, class P>
__global__ void __launch_bounds__(T) k_segsort(SegSortArgs A, int tier, P prod)
{
    extern __shared__ uint64_t seg_lds[];
    const int tid = threadIdx.x;
    const int cnt = A.tier_cnt[tier];
    const int32_t *list = A.tier_list + (size_t)tier * A.nseg;
    for (int it = blockIdx.x; it < cnt; it += gridDim.x) {
        const int s = list[it];
        const int32_t d0 = A.seg_beg[s];
        const int n = A.seg_end[s] - d0;
        uint64_t *dst = A.out + d0;
        // the keys into LDS, pads behind the segment
        const int npad = ((n + E - 1) / E) * E;        // the threads that own a real key own E slots
        prod.fill(A, s, seg_lds, n, npad, tid, T);
        __syncthreads();
        uint64_t k[E];
        const bool live = tid * E < n;                // this thread owns at least one real key
        if (live) {
#pragma unroll
            for (int e = 0; e < E; ++e) k[e] = seg_lds[tid * E + e];
            d_regsort<E>(k);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) k[e] = SEGSORT_PAD;
        }
        // merge rounds: runs of r keys (tpr threads each) are merged pairwise until one run covers the segment
        for (int r = E, tpr = 1; r < n; r <<= 1, tpr <<= 1) {
            __syncthreads();                          // everyone has read what the previous round left in LDS
            if (live) {
#pragma unroll
                for (int e = 0; e < E; ++e) seg_lds[tid * E + e] = k[e];
            }
            __syncthreads();
            const int base = (tid / (2 * tpr)) * 2 * r;
            if (base < n && live) {                   // (a dead thread's outputs are pads: its registers already hold them)
                // the pair's real extent: run A = [base, base + la), run B = [base + r, base + r + lb); slots beyond npad were never written
                const int la = min(r, npad - base), lb = max(0, min(r, npad - base - r));
                const uint64_t *RA = seg_lds + base, *RB = seg_lds + base + r;
                const int d = (tid & (2 * tpr - 1)) * E;
                int lo = max(0, d - lb), hi = min(d, la);
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (RA[mid] <= RB[d - 1 - mid]) lo = mid + 1; else hi = mid;
                }
                int a = lo, b = d - lo;
                uint64_t ka = a < la ? RA[a] : SEGSORT_PAD, kb = b < lb ? RB[b] : SEGSORT_PAD;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool ta = ka <= kb;
                    k[e] = ta ? ka : kb;
                    if (ta) { ++a; ka = a < la ? RA[a] : SEGSORT_PAD; }
                    else { ++b; kb = b < lb ? RB[b] : SEGSORT_PAD; }
                }
            }
        }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int e = 0; e < E; ++e) seg_lds[tid * E + e] = k[e];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = tid + e * T;
            if (i < n) dst[i] = seg_lds[i];
        }
        __syncthreads();                              // the next segment's load overwrites LDS
    }
}
