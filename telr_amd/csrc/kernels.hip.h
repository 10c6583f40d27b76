// kernels.hip.h — gfx950 device code of the TELR alignment engine.
//
// Everything here is integer / byte work bounded by HBM and LDS traffic (no MFMA by
// design: scoring is integer DP, not a dense contraction).  Data layout:
//   * sequences: 2 bits/base in uint32 words (16 bases per word, base p at bits 2*(p&15)),
//     plus an ambiguity bitmask (1 bit/base, 32 per word); every sequence starts at a
//     64-base (16-byte) boundary so tiles are read with aligned, coalesced loads;
//   * minimizers: SoA  x (u64 hash<<8|span)  /  y (u32 pos<<1|strand);
//   * index: distinct hashes (u64, ascending) + u32 offsets into a u32 position array (what the oracle is compared
//     with); the seeding kernel probes an open-addressing table of 16-byte slots {hash, first occurrence, count} in
//     front of a 2-bit-per-slot home bitmap (one 64-byte line per probe);
//   * anchors: ONE sortable u64 per anchor (strand | global target pos | query pos | span),
//     half the footprint of the classic 16-byte anchor, so seeding, sorting and chaining
//     each move 8 B per anchor;
//   * chaining: forward push in registers (one wave per query, v_readlane broadcasts); back-tracking: owner / depth
//     sweeps with pointer jumping (no walker); pass-1 chain selection: one wave per query;
//   * DP: anti-diagonal sweep.  Gap fills with <= 128 diagonals run in packed int16 registers (two diagonals per
//     32-bit register, one problem per lane for bands <= 32: d_dp_pkr), wider ones in packed multi-wave or int32
//     register kernels, only bands > 1024 keep the DP state in LDS; one trace-back byte per cell streamed to an HBM
//     scratch (wave-interleaved 512-byte stores for the lane-per-problem classes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TELR_NEG      (-(1 << 28))
#define TELR_TPAD     16384
#define SK_TILE       1024          // minimizer slots per sketch tile
#define SK_THREADS    256
#define DP_DMAX       4096          // widest band (diagonals) the DP kernel accepts

struct ChainOpt {            // subset of telr_map_opt the device needs
    int32_t max_gap, bw, min_cnt, min_chain_score, chain_gap_q8, chain_skip_q8;
    int32_t dense_n, dense_span, mw_n;    // k_chain's choice of loop per run (a speed matter only: every loop yields the same f and p), see d_chain_dense
};
struct DpOpt {
    int32_t a, b, q, e, q2, e2, sc_ambi, zdrop;
    // convex gap cost (telr_map_opt.cx_*; cx_scale > 0): a, b, sc_ambi, zdrop above are then already multiplied by cx_scale, q / e / q2 / e2
    // are unused, and the E2 / F2 slots of every kernel carry the LENGTH of the gap its E / F state ends with (capped at cx_flat, from
    // where the extension is flat): ext(len) = max(cx_emin, cx_emax - cx_dec * len)
    int32_t cx_scale, cx_open, cx_emax, cx_emin, cx_dec, cx_flat;
};
__host__ __device__ __forceinline__ int d_cx_ext(const DpOpt &o, int len) { int v = o.cx_emax - o.cx_dec * len; return v > o.cx_emin ? v : o.cx_emin; }
__host__ __device__ __forceinline__ int d_cx_cost(const DpOpt &o, int L)          // a whole gap of L bases
{
    int t = o.cx_open;
    for (int i = 0; i < L && i < o.cx_flat; ++i) t += d_cx_ext(o, i);
    if (L > o.cx_flat) t += (L - o.cx_flat) * o.cx_emin;
    return t;
}

// ---------------------------------------------------------------------------------------
// small device helpers
__device__ __forceinline__ uint64_t d_hash64(uint64_t key, uint64_t mask)
{
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

// nb (<=28) bases starting at base index b, first base in the LOW bits
__device__ __forceinline__ uint64_t d_get_bases(const uint32_t *__restrict__ seq2, int64_t b, int nb)
{
    int64_t w = b >> 4; int sh = (int)(b & 15) * 2;
    uint64_t lo = (uint64_t)seq2[w] | (uint64_t)seq2[w + 1] << 32;
    uint64_t hi = seq2[w + 2];
    uint64_t v = lo >> sh;
    if (sh) v |= hi << (64 - sh);
    return v & ((1ULL << 2 * nb) - 1);
}
__device__ __forceinline__ uint32_t d_get_nbits(const uint32_t *__restrict__ nm, int64_t b, int nb)
{
    int64_t w = b >> 5; int sh = (int)(b & 31);
    uint64_t v = ((uint64_t)nm[w] | (uint64_t)nm[w + 1] << 32) >> sh;
    return (uint32_t)(v & ((1ULL << nb) - 1));
}
__device__ __forceinline__ int d_base(const uint32_t *__restrict__ seq2, const uint32_t *__restrict__ nm, int64_t b)
{
    int c = (seq2[b >> 4] >> ((int)(b & 15) * 2)) & 3;
    return ((nm[b >> 5] >> (int)(b & 31)) & 1) ? 4 : c;
}
// reverse the order of the 2-bit groups of the low 2k bits
__device__ __forceinline__ uint64_t d_rev2(uint64_t v, int k)
{
    uint64_t r = __brevll(v) >> (64 - 2 * k);
    return ((r & 0x5555555555555555ULL) << 1) | ((r >> 1) & 0x5555555555555555ULL);
}

// ---------------------------------------------------------------------------------------
// 1. minimizer sketch (non-HPC): one block per tile of SK_TILE slots.
//    mode 0: count selected slots per tile; mode 1: write (x,y) at tile_off[t].
//    Selection rule (oracle sketch()): slot u selected iff valid and the run of slots
//    around u whose x >= x_u is at least min(w, nslots) long.
struct SketchArgs {
    const uint32_t *seq2, *nmask;
    const int64_t *boff;       // base offset of each sequence in the packed arrays
    const int32_t *len;
    const uint32_t *goff;      // global coordinate offsets (index build) or nullptr (queries: 0)
    const int32_t *tile_seq, *tile_u0;
    int32_t k, w;
    int32_t *tile_cnt;         // mode 0 out
    const int32_t *tile_off;   // mode 1 in (exclusive scan of tile_cnt)
    uint64_t *out_x; uint32_t *out_y;
};

// Is slot u (value x at xs[s]) a minimizer?  The run of slots with value >= x around u, both sides, capped at the sequence
// ends, must reach `need` slots.  Every lane tests all `halo` neighbours (a lane that stops at its first smaller neighbour
// would still wait for the wave's slowest lane); HALO > 0 = halo known at compile time, fully unrolled.
template <int HALO, typename T>
__device__ __forceinline__ bool d_mz_selected(const T *xs, int s, int u, int ns, int need, int halo, T x)
{
    int Lc, Rc;
    if (HALO) {
        int lf = HALO + 1, rf = HALO + 1;
#pragma unroll
        for (int t = HALO; t >= 1; --t) { if (xs[s - t] < x) lf = t; if (xs[s + t] < x) rf = t; }
        const int lmax = u < HALO ? u : HALO, rmax = ns - 1 - u < HALO ? ns - 1 - u : HALO;
        Lc = lf - 1 < lmax ? lf - 1 : lmax; Rc = rf - 1 < rmax ? rf - 1 : rmax;
    } else if (halo <= 31) {
        uint32_t lm = 0, rm = 0;
        for (int t = 1; t <= halo; ++t) {
            lm |= (uint32_t)(u - t >= 0 && xs[s - t] >= x) << (t - 1);
            rm |= (uint32_t)(u + t < ns && xs[s + t] >= x) << (t - 1);
        }
        Lc = __ffs((int)~lm) - 1; Rc = __ffs((int)~rm) - 1;
    } else {                                  // very wide windows: plain scans
        Lc = 0; Rc = 0;
        while (Lc < halo && u - Lc - 1 >= 0 && xs[s - Lc - 1] >= x) ++Lc;
        while (Rc < halo && u + Rc + 1 < ns && xs[s + Rc + 1] >= x) ++Rc;
    }
    return Lc + Rc + 1 >= need;
}

template <int MODE>
__global__ void __launch_bounds__(SK_THREADS) k_sketch(SketchArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int halo = A.w - 1, nslot = SK_TILE + 2 * halo;
    uint64_t *xs = (uint64_t*)smem;
    uint8_t *zs = (uint8_t*)(xs + nslot);
    __shared__ int32_t wsum[SK_THREADS / 64];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int sid = A.tile_seq[t], u0 = A.tile_u0[t];
    const int L = A.len[sid], ns = L - A.k + 1, k = A.k;
    const int64_t base = A.boff[sid];
    const uint64_t mask = (1ULL << 2 * k) - 1;
    for (int s = tid; s < nslot; s += SK_THREADS) {
        int u = u0 - halo + s;
        uint64_t x = UINT64_MAX; uint8_t z = 0;
        if (u >= 0 && u < ns) {
            uint32_t nb = d_get_nbits(A.nmask, base + u, k);
            if (nb == 0) {
                uint64_t v = d_get_bases(A.seq2, base + u, k);
                uint64_t fw = d_rev2(v, k), rv = (~v) & mask;
                if (fw != rv) {
                    z = fw < rv ? 0 : 1;
                    x = d_hash64(z ? rv : fw, mask) << 8 | (uint64_t)k;
                }
            }
        }
        xs[s] = x; zs[s] = z;
    }
    __syncthreads();
    const int need = A.w < ns ? A.w : ns;
    uint32_t sel = 0;
#pragma unroll
    for (int c = 0; c < SK_TILE / SK_THREADS; ++c) {
        int s = halo + tid * (SK_TILE / SK_THREADS) + c, u = u0 - halo + s;
        uint64_t x = xs[s];
        const bool is_mz = halo == 9 ? d_mz_selected<9, uint64_t>(xs, s, u, ns, need, halo, x) : d_mz_selected<0, uint64_t>(xs, s, u, ns, need, halo, x);
        if (u < ns && x != UINT64_MAX && is_mz) sel |= 1u << c;
    }
    int cnt = __popc(sel);
    // block exclusive scan of cnt (wave scan + cross-wave in LDS)
    int lane = tid & 63, wv = tid >> 6, inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int i = 0; i < SK_THREADS / 64; ++i) { if (i < wv) wbase += wsum[i]; total += wsum[i]; }
    if (MODE == 0 || MODE == 2) {
        if (tid == 0) A.tile_cnt[t] = total;
    }
    if (MODE != 0) {
        int64_t o = (MODE == 2 ? (int64_t)t * SK_TILE : (int64_t)A.tile_off[t]) + wbase + inc - cnt;
        uint32_t g0 = A.goff ? A.goff[sid] : 0u;
#pragma unroll
        for (int c = 0; c < SK_TILE / SK_THREADS; ++c) if (sel >> c & 1) {
            int s = halo + tid * (SK_TILE / SK_THREADS) + c, u = u0 - halo + s;
            A.out_x[o] = xs[s];
            A.out_y[o] = (g0 + (uint32_t)(u + k - 1)) << 1 | zs[s];
            ++o;
        }
    }
}

// k <= 15: k-mer, hash and the LDS window all fit in 32 bits (same values as the 64-bit path, checked step by step:
// every masked step of d_hash64 only depends on the low 2k bits)
__device__ __forceinline__ uint32_t d_hash32(uint32_t key, uint32_t mask)
{
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}
// One k-mer slot (k <= 15) -> its LDS word: hash + 1 (so that 0 is free to mean "no such window" below), SK_NONE for a slot
// outside the sequence, with an N, or equal to its reverse complement.  v = the k bases, first base in the low bits.
#define SK_NONE 0xffffffffu
__device__ __forceinline__ uint32_t d_slot32(uint32_t v, bool bad, int k, uint32_t mask, uint32_t &z)
{
    const uint32_t r = __brev(v) >> (32 - 2 * k);
    const uint32_t fw = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u), rv = (~v) & mask;
    z = fw < rv ? 0u : 1u;
    const uint32_t x = d_hash32(fw < rv ? fw : rv, mask) + 1u;
    return (bad || fw == rv) ? SK_NONE : x;
}
// w = 10: which of the thread's 4 consecutive slots are minimizers.  Slot u is one iff it is the minimum of SOME window of 10
// slots, i.e. iff  max over the 10 windows holding u of (the window's minimum)  equals its own value: a sliding minimum followed
// by a sliding maximum over values held in registers (min3 / max3 trees, ~17 operations per slot instead of 18 LDS reads and
// compare-selects).  p = the 22 LDS words from 9 slots before the first own slot; window a (0..12) starts at slot ua0 + a and
// only counts when it lies inside the sequence (edge tiles).
__device__ __forceinline__ uint32_t d_mz_sel4_w10(const uint32_t *p, int ua0, int ns, bool edge)
{
    uint32_t v[24];
#pragma unroll
    for (int q = 0; q < 6; ++q) { const uint4 t = ((const uint4*)p)[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
    uint32_t t3[19], m[13], T3[10];
#pragma unroll
    for (int i = 0; i < 19; ++i) { const uint32_t a = v[i] < v[i + 1] ? v[i] : v[i + 1]; t3[i] = a < v[i + 2] ? a : v[i + 2]; }
#pragma unroll
    for (int a = 0; a < 13; ++a) {
        uint32_t x = t3[a] < t3[a + 3] ? t3[a] : t3[a + 3]; x = x < t3[a + 6] ? x : t3[a + 6];
        m[a] = x < v[a + 9] ? x : v[a + 9];
    }
    if (edge) {
#pragma unroll
        for (int a = 0; a < 13; ++a) if ((uint32_t)(ua0 + a) > (uint32_t)(ns - 10)) m[a] = 0;
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) { const uint32_t a = m[i] > m[i + 1] ? m[i] : m[i + 1]; T3[i] = a > m[i + 2] ? a : m[i + 2]; }
    uint32_t sel = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint32_t M = T3[c] > T3[c + 3] ? T3[c] : T3[c + 3]; M = M > T3[c + 6] ? M : T3[c + 6]; M = M > m[c + 9] ? M : m[c + 9];
        if (v[9 + c] != SK_NONE && M == v[9 + c]) sel |= 1u << c;
    }
    return sel;
}

// The same for w = 5 (the ngmlr-* presets): p = the 12 LDS words from 4 slots before the first own slot, window a (0..7) starts at
// slot ua0 + a.  (The neighbour scans of the general path read LDS with a stride of four words per lane: 73 % of its LDS cycles
// were bank conflicts -- profiles/r04: SQ_LDS_BANK_CONFLICT of k_sketch32<2, 0>.)
__device__ __forceinline__ uint32_t d_mz_sel4_w5(const uint32_t *p, int ua0, int ns, bool edge)
{
    uint32_t v[12];
#pragma unroll
    for (int q = 0; q < 3; ++q) { const uint4 t = ((const uint4*)p)[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
    uint32_t t2[11], m[8];
#pragma unroll
    for (int i = 0; i < 11; ++i) t2[i] = v[i] < v[i + 1] ? v[i] : v[i + 1];
#pragma unroll
    for (int a = 0; a < 8; ++a) { uint32_t x = t2[a] < t2[a + 2] ? t2[a] : t2[a + 2]; m[a] = x < v[a + 4] ? x : v[a + 4]; }
    if (edge) {
#pragma unroll
        for (int a = 0; a < 8; ++a) if ((uint32_t)(ua0 + a) > (uint32_t)(ns - 5)) m[a] = 0;
    }
    uint32_t T2[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) T2[i] = m[i] > m[i + 1] ? m[i] : m[i + 1];
    uint32_t sel = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint32_t M = T2[c] > T2[c + 2] ? T2[c] : T2[c + 2]; M = M > m[c + 4] ? M : m[c + 4];
        if (v[4 + c] != SK_NONE && M == v[4 + c]) sel |= 1u << c;
    }
    return sel;
}

// Thread t of a tile owns the 4 consecutive slots u0 + 4t .. +3: it hashes them from one 64-bit window of the packed bases
// (strand bits stay in registers), the first 2*halo threads also hash one halo slot each, and after the barrier every thread
// tests its own 4 slots.  The selected slots are ranked with ballots (no shuffles).
template <int MODE, int HALO>        // HALO = 9 / 4: w = 10 / 5, the register formulations above; 0 = any w, neighbour scans in LDS
__global__ void __launch_bounds__(SK_THREADS) k_sketch32(SketchArgs A)
{
    static_assert(SK_TILE == 4 * SK_THREADS, "4 slots per thread");
    extern __shared__ __align__(16) unsigned char smem[];
    const int halo = HALO ? HALO : A.w - 1;
    uint32_t *xs = (uint32_t*)smem;
    __shared__ int32_t wsum[SK_THREADS / 64];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int sid = A.tile_seq[t], u0 = A.tile_u0[t];
    const int L = A.len[sid], ns = L - A.k + 1, k = A.k;
    const int64_t base = A.boff[sid];
    const uint32_t mask = (1u << 2 * k) - 1, kmask = (1u << k) - 1;
    const int uo = u0 + tid * 4;                       // first own slot
    uint32_t xo[4] = {SK_NONE, SK_NONE, SK_NONE, SK_NONE}, zb = 0;
    if (uo < ns) {
        const int64_t bb = base + uo; const int64_t wi = bb >> 4; const int sh = (int)(bb & 15) * 2;
        const uint32_t w0 = A.seq2[wi], w1 = A.seq2[wi + 1], w2 = A.seq2[wi + 2];
        const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);   // bases uo .. uo+31
        const int64_t nw = bb >> 5;
        const uint32_t nb = __builtin_amdgcn_alignbit(A.nmask[nw + 1], A.nmask[nw], (int)(bb & 31));            // their N flags
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t z;
            xo[c] = d_slot32(__builtin_amdgcn_alignbit(hi, lo, 2 * c) & mask, ((nb >> c) & kmask) != 0 || uo + c >= ns, k, mask, z);
            zb |= z << c;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) xs[halo + tid * 4 + c] = xo[c];
    for (int h = tid; h < 2 * halo; h += SK_THREADS) { // halo slots: u0-halo .. u0-1 and u0+SK_TILE .. +halo-1
        const int s = h < halo ? h : SK_TILE + h, u = u0 - halo + s;
        uint32_t x = SK_NONE;
        if (u >= 0 && u < ns) {
            const int64_t bb = base + u; const int64_t wi = bb >> 4;
            const uint32_t v = __builtin_amdgcn_alignbit(A.seq2[wi + 1], A.seq2[wi], (int)(bb & 15) * 2) & mask;
            uint32_t z;
            x = d_slot32(v, d_get_nbits(A.nmask, bb, k) != 0, k, mask, z);
        }
        xs[s] = x;
    }
    __syncthreads();
    uint32_t sel = 0;
    if (HALO == 9 && ns >= 10) {
        const bool edge = u0 - 9 < 0 || u0 + SK_TILE + 9 > ns;
        sel = d_mz_sel4_w10(xs + tid * 4, uo - 9, ns, edge);
    } else if (HALO == 4 && ns >= 5) {
        const bool edge = u0 - 4 < 0 || u0 + SK_TILE + 4 > ns;
        sel = d_mz_sel4_w5(xs + tid * 4, uo - 4, ns, edge);
    } else {
        const int need = A.w < ns ? A.w : ns;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (uo + c < ns && xo[c] != SK_NONE && d_mz_selected<0, uint32_t>(xs, halo + tid * 4 + c, uo + c, ns, need, halo, xo[c])) sel |= 1u << c;
    }
    // rank of the thread's first selected slot inside the tile: lanes below in the wave (ballots) + waves below (LDS)
    const int cnt = __popc(sel), wv = tid >> 6;
    int below = 0, wtot = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint64_t bal = __ballot((sel >> c) & 1u);
        below += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        wtot += __popcll(bal);
    }
    if ((tid & 63) == 0) wsum[wv] = wtot;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int i = 0; i < SK_THREADS / 64; ++i) { if (i < wv) wbase += wsum[i]; total += wsum[i]; }
    if (MODE == 0 || MODE == 2) {
        if (tid == 0) A.tile_cnt[t] = total;
    }
    if (MODE != 0 && cnt) {
        int64_t o = (MODE == 2 ? (int64_t)t * SK_TILE : (int64_t)A.tile_off[t]) + wbase + below;
        const uint32_t g0 = A.goff ? A.goff[sid] : 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (sel >> c & 1) {
            A.out_x[o] = (uint64_t)(xo[c] - 1u) << 8 | (uint64_t)k;
            A.out_y[o] = (g0 + (uint32_t)(uo + c + k - 1)) << 1 | ((zb >> c) & 1u);
            ++o;
        }
    }
}

// staging -> dense: one block per tile copies its tile_cnt entries to tile_off (coalesced both ways)
__global__ void __launch_bounds__(256) k_sketch_compact(const uint64_t *__restrict__ sx, const uint32_t *__restrict__ sy, const int32_t *__restrict__ tile_cnt,
                                                        const int32_t *__restrict__ tile_off, uint64_t *__restrict__ out_x, uint32_t *__restrict__ out_y)
{
    const int t = blockIdx.x, n = tile_cnt[t];
    const int64_t src = (int64_t)t * SK_TILE, dst = tile_off[t];
    for (int i = threadIdx.x; i < n; i += blockDim.x) { out_x[dst + i] = sx[src + i]; out_y[dst + i] = sy[src + i]; }
}

// ---------------------------------------------------------------------------------------
// 1b. homopolymer-compressed (HPC) sketch, map-pb.  A pre-pass compacts every sequence into its runs:
//     hcode[h] = base code (0-4) of run h, hstart[h] = first position of the run; then the sketch runs
//     over run indices: k consecutive runs form a k-mer, span = last base of run u+k-1 - first base of
//     run u + 1 (k-mers with span >= 256 are invalid), position = last base of the last run.
__device__ __forceinline__ int d_seq_of_chunk(const int64_t *__restrict__ boff, int n, int64_t b)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (boff[mid] <= b) lo = mid; else hi = mid - 1; }
    return lo;
}
// one thread per 64-base chunk (sequences start on 64-base boundaries): run-start flags and their count
// A 64-base chunk starts on a word boundary (sequences are padded to 64 bases): its 4 base words and 2 mask words are
// loaded once; base x of the chunk = code (2 bits) or 4 when the mask bit is set.
struct Chunk64 { uint32_t w[4]; uint32_t m[2]; };
__device__ __forceinline__ Chunk64 d_chunk_load(const uint32_t *__restrict__ seq2, const uint32_t *__restrict__ nmask, int64_t b0)
{
    Chunk64 c;
    const uint4 v = *(const uint4*)(seq2 + (b0 >> 4));
    c.w[0] = v.x; c.w[1] = v.y; c.w[2] = v.z; c.w[3] = v.w;
    const uint2 n = *(const uint2*)(nmask + (b0 >> 5));
    c.m[0] = n.x; c.m[1] = n.y;
    return c;
}
__device__ __forceinline__ int d_chunk_base(const Chunk64 &c, int x)
{
    const uint64_t lo = (uint64_t)c.w[1] << 32 | c.w[0], hi = (uint64_t)c.w[3] << 32 | c.w[2];
    const uint64_t mm = (uint64_t)c.m[1] << 32 | c.m[0];
    const int code = (int)(((x < 32 ? lo : hi) >> ((x & 31) * 2)) & 3u);
    return ((mm >> x) & 1u) ? 4 : code;
}
__global__ void k_hpc_flags(const uint32_t *__restrict__ seq2, const uint32_t *__restrict__ nmask, const int64_t *__restrict__ boff,
                            const int32_t *__restrict__ len, int32_t sid0, int32_t nseq, int64_t chunk0, int32_t nchunk,
                            uint64_t *__restrict__ flags, int32_t *__restrict__ cnt)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunk) return;
    const int64_t b0 = (chunk0 + c) * 64;
    const int sid = sid0 + d_seq_of_chunk(boff + sid0, nseq, b0);
    const int64_t sb = boff[sid];
    const int L = len[sid];
    const int p0 = (int)(b0 - sb);
    const Chunk64 C = d_chunk_load(seq2, nmask, b0);
    uint64_t f = 0;
    int prev = p0 > 0 ? d_base(seq2, nmask, b0 - 1) : -1;
    const int nx = L - p0 < 64 ? L - p0 : 64;
    for (int x = 0; x < nx; ++x) {
        const int cur = d_chunk_base(C, x);
        if (cur != prev) f |= 1ULL << x;
        prev = cur;
    }
    flags[c] = f; cnt[c] = __popcll(f);
}
// Runs of 128 chunks per block: every thread drops the runs of its chunk into LDS at their rank inside the block, then the
// block writes codes and start positions out with consecutive threads on consecutive elements.
__global__ void __launch_bounds__(128) k_hpc_scatter(const uint32_t *__restrict__ seq2, const uint32_t *__restrict__ nmask, const int64_t *__restrict__ boff,
                                                     int32_t sid0, int32_t nseq, int64_t chunk0, int32_t nchunk, const uint64_t *__restrict__ flags,
                                                     const int32_t *__restrict__ coff, uint8_t *__restrict__ hcode, uint32_t *__restrict__ hstart)
{
    __shared__ uint32_t s_start[128 * 64];
    __shared__ uint8_t s_code[128 * 64];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int cfirst = blockIdx.x * blockDim.x, clast = cfirst + 128 < nchunk ? cfirst + 128 : nchunk;
    const int o_first = coff[cfirst], o_end = coff[clast];
    if (c < nchunk) {
        const int64_t b0 = (chunk0 + c) * 64;
        const int sid = sid0 + d_seq_of_chunk(boff + sid0, nseq, b0);
        const int p0 = (int)(b0 - boff[sid]);
        const Chunk64 C = d_chunk_load(seq2, nmask, b0);
        uint64_t f = flags[c]; int o = coff[c] - o_first;
        while (f) {
            const int x = __ffsll((long long)f) - 1; f &= f - 1;
            s_code[o] = (uint8_t)d_chunk_base(C, x); s_start[o] = (uint32_t)(p0 + x); ++o;
        }
    }
    __syncthreads();
    const int n = o_end - o_first;
    for (int z = threadIdx.x; z < n; z += 128) { hstart[o_first + z] = s_start[z]; hcode[o_first + z] = s_code[z]; }
}
// per sequence: offset / count of its runs = scan value at its first chunk
__global__ void k_hpc_seq_offsets(const int64_t *__restrict__ boff, int32_t sid0, int32_t nseq, int64_t chunk0, const int32_t *__restrict__ coff,
                                  int32_t nchunk, int32_t total, int32_t *__restrict__ hoff /* [nseq+1] */)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > nseq) return;
    if (s == nseq) { hoff[s] = total; return; }
    const int64_t c = boff[sid0 + s] / 64 - chunk0;
    hoff[s] = c < nchunk ? coff[c] : total;
}

struct SketchHpcArgs {
    const uint8_t *hcode; const uint32_t *hstart;
    const int32_t *hoff;        // [nseq+1] run offsets, local sequence index
    const int32_t *len;         // sequence lengths, local sequence index
    const uint32_t *goff;       // or nullptr
    const int32_t *tile_seq, *tile_u0;
    int32_t k, w;
    int32_t *tile_cnt; const int32_t *tile_off;
    uint64_t *out_x; uint32_t *out_y;
};

template <int MODE>
__global__ void __launch_bounds__(SK_THREADS) k_sketch_hpc(SketchHpcArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int halo = A.w - 1, nslot = SK_TILE + 2 * halo, k = A.k;
    uint64_t *xs = (uint64_t*)smem;
    uint32_t *ps = (uint32_t*)(xs + nslot);             // position<<1 | strand of each slot
    uint8_t *cb = (uint8_t*)(ps + nslot);               // run codes of the tile: nslot + k - 1 bytes
    uint32_t *cw = (uint32_t*)(cb + ((nslot + k - 1 + 3) & ~3)), *cn = cw + (nslot + k - 1 + 15) / 16 + 3;
    __shared__ int32_t wsum[SK_THREADS / 64];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int sid = A.tile_seq[t], u0 = A.tile_u0[t];
    const int h0 = A.hoff[sid], nh = A.hoff[sid + 1] - h0, ns = nh - k + 1, L = A.len[sid];
    const uint64_t mask = (1ULL << 2 * k) - 1;
    const int lo = u0 - halo;
    for (int s = tid; s < nslot + k - 1; s += SK_THREADS) { int r = lo + s; cb[s] = (r >= 0 && r < nh) ? A.hcode[h0 + r] : 4; }
    __syncthreads();
    // 2-bit packed copy of the codes + ambiguity bits (k-mers are then two shifts instead of a loop over k bytes)
    const int ncode = nslot + k - 1;
    for (int wq = tid; wq < (ncode + 15) / 16 + 3; wq += SK_THREADS) {
        uint32_t v = 0;
        for (int z = 0; z < 16; ++z) { const int q = wq * 16 + z; if (q < ncode) v |= (uint32_t)(cb[q] & 3) << (2 * z); }
        cw[wq] = v;
    }
    for (int wq = tid; wq < (ncode + 31) / 32 + 2; wq += SK_THREADS) {
        uint32_t v = 0;
        for (int z = 0; z < 32; ++z) { const int q = wq * 32 + z; if (q >= ncode || cb[q] > 3) v |= 1u << z; }
        cn[wq] = v;
    }
    __syncthreads();
    for (int s = tid; s < nslot; s += SK_THREADS) {
        const int u = lo + s;
        uint64_t x = UINT64_MAX; uint32_t pz = 0;
        if (u >= 0 && u < ns) {
            // k run codes from the 2-bit packed copy (first run in the low bits), as the plain sketch reads bases
            const int wi = s >> 4, sh = (s & 15) * 2;
            const uint64_t lo64 = (uint64_t)cw[wi] | (uint64_t)cw[wi + 1] << 32, hi64 = cw[wi + 2];
            uint64_t v = lo64 >> sh;
            if (sh) v |= hi64 << (64 - sh);
            v &= mask;
            const uint64_t nb = (((uint64_t)cn[s >> 5] | (uint64_t)cn[(s >> 5) + 1] << 32) >> (s & 31)) & ((1ULL << k) - 1);
            const bool ok = nb == 0;
            const uint64_t fw = d_rev2(v, k), rv = (~v) & mask;
            const uint32_t first = A.hstart[h0 + u];
            const uint32_t last = (u + k < nh ? A.hstart[h0 + u + k] : (uint32_t)L) - 1;     // last base of run u+k-1
            const uint32_t span = last - first + 1;
            if (ok && fw != rv && span < 256) {
                const uint32_t z = fw < rv ? 0 : 1;
                x = d_hash64(z ? rv : fw, mask) << 8 | (uint64_t)span;
                pz = last << 1 | z;
            }
        }
        xs[s] = x; ps[s] = pz;
    }
    __syncthreads();
    const int need = A.w < ns ? A.w : ns;
    uint32_t sel = 0;
#pragma unroll
    for (int c = 0; c < SK_TILE / SK_THREADS; ++c) {
        int s = halo + tid * (SK_TILE / SK_THREADS) + c, u = lo + s;
        uint64_t x = xs[s];
        const bool is_mz = halo == 9 ? d_mz_selected<9, uint64_t>(xs, s, u, ns, need, halo, x) : d_mz_selected<0, uint64_t>(xs, s, u, ns, need, halo, x);
        if (u < ns && x != UINT64_MAX && is_mz) sel |= 1u << c;
    }
    int cnt = __popc(sel);
    int lane = tid & 63, wv = tid >> 6, inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int i = 0; i < SK_THREADS / 64; ++i) { if (i < wv) wbase += wsum[i]; total += wsum[i]; }
    if (MODE == 0 || MODE == 2) {
        if (tid == 0) A.tile_cnt[t] = total;
    }
    if (MODE != 0) {
        int64_t o = (MODE == 2 ? (int64_t)t * SK_TILE : (int64_t)A.tile_off[t]) + wbase + inc - cnt;
        const uint32_t g0 = A.goff ? A.goff[sid] : 0u;
#pragma unroll
        for (int c = 0; c < SK_TILE / SK_THREADS; ++c) if (sel >> c & 1) {
            int s = halo + tid * (SK_TILE / SK_THREADS) + c;
            A.out_x[o] = xs[s];
            A.out_y[o] = ((g0 << 1) + ps[s]);
            ++o;
        }
    }
}

// ---------------------------------------------------------------------------------------
// 2. index build helpers
// index build: the sort keys (hash = x >> 8) and their values (y) into the arrays the radix sort starts from
__global__ void __launch_bounds__(256) k_ix_hash_keys(const uint64_t *__restrict__ x, const uint32_t *__restrict__ y, int64_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = x[i] >> 8;
    if (val != y) val[i] = y[i];
}
__global__ void k_head_flags(const uint64_t *__restrict__ h, int64_t n, int32_t *__restrict__ flag)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || h[i] != h[i - 1]) ? 1 : 0;
}
__global__ void k_write_entries(const uint64_t *__restrict__ h, const int32_t *__restrict__ flag, const int32_t *__restrict__ rank,
                                int64_t n, uint64_t *__restrict__ ent_hash, uint32_t *__restrict__ ent_off, int32_t n_ent)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) { ent_hash[rank[i]] = h[i]; ent_off[rank[i]] = (uint32_t)i; }
    if (i == 0) ent_off[n_ent] = (uint32_t)n;
}
__global__ void k_ent_counts(const uint32_t *__restrict__ ent_off, int32_t n_ent, uint32_t *__restrict__ cnt)
{
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_ent) cnt[i] = ent_off[i + 1] - ent_off[i];
}
// probe table: open addressing (linear probing, load <= 1/2) over 16-byte slots {hash, first occurrence, count}: one
// 64-byte line answers almost every probe, where bucket table -> hash array -> offset array needs three.  The slot
// comes from a multiplicative re-hash: minimizer hashes are window MINIMA, i.e. heavily skewed towards small values,
// so their top bits alone would pile ten times the average load onto the low slots.
__device__ __forceinline__ uint32_t d_ht_slot(uint64_t h, int ht_shift, uint32_t ht_mask)
{
    return (uint32_t)((h * 0x9E3779B97F4A7C15ULL) >> ht_shift) & ht_mask;
}
#ifndef HT_FB_LOG
#define HT_FB_LOG 1          /* filter bits per table slot = 2^HT_FB_LOG */
#endif
__device__ __forceinline__ uint32_t d_ht_filter_bit(uint64_t h, int ht_shift) { return (uint32_t)((h * 0x9E3779B97F4A7C15ULL) >> (ht_shift - HT_FB_LOG)); }
struct HtSlot { uint64_t hash; uint32_t off, cnt; };
#define HT_EMPTY 0xffffffffffffffffULL
struct IndexView {
    const uint64_t *ent_hash; const uint32_t *ent_off; const uint32_t *pos; const uint32_t *bstart;
    const uint32_t *goff; const int32_t *tlen;
    int32_t n_ent; int32_t shift; int32_t k, w;
    const HtSlot *ht; int32_t ht_shift; uint32_t ht_mask;
    const uint32_t *ht_home;     // HT_FB bits per slot: some minimizer's re-hash starts with these log2(slots) + log2(HT_FB) bits
};
// The filter bitmap is 1/64 of the table (2 MB for a 23-Mb genome: L2-resident): four of five read minimizers carry a
// sequencing error and are not in the index at all, and ~3/4 of those find their bit clear, which answers the probe
// without fetching a table line from HBM.
__global__ void k_ht_build(const uint64_t *__restrict__ ent_hash, const uint32_t *__restrict__ ent_off, int32_t n_ent, int ht_shift, uint32_t ht_mask, HtSlot *__restrict__ ht,
                           uint32_t *__restrict__ ht_home, const uint32_t *__restrict__ pos)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_ent) return;
    const uint64_t h = ent_hash[e];
    uint32_t s = d_ht_slot(h, ht_shift, ht_mask);
    { const uint32_t f = d_ht_filter_bit(h, ht_shift); atomicOr(&ht_home[f >> 5], 1u << (f & 31)); }
    for (;;) {
        const unsigned long long old = atomicCAS((unsigned long long*)&ht[s].hash, (unsigned long long)HT_EMPTY, (unsigned long long)h);
        // a minimizer with ONE occurrence (most of them) carries the occurrence itself instead of its place in `pos`: the seeding
        // kernels then need no second scattered read for it
        if (old == HT_EMPTY) { const uint32_t c = ent_off[e + 1] - ent_off[e]; ht[s].off = c == 1 ? pos[ent_off[e]] : ent_off[e]; ht[s].cnt = c; return; }
        s = (s + 1) & ht_mask;
    }
}
// -> true and (off, cnt) of the minimizer's occurrence list, or false
template <bool FILTER = true>
__device__ __forceinline__ bool d_ht_lookup(const IndexView &I, uint64_t h, uint32_t &off, uint32_t &cnt)
{
    uint32_t s = d_ht_slot(h, I.ht_shift, I.ht_mask);
    if (FILTER) { const uint32_t f = d_ht_filter_bit(h, I.ht_shift); if (!((I.ht_home[f >> 5] >> (f & 31)) & 1u)) return false; }
    for (;;) {
        const uint4 v = *(const uint4*)&I.ht[s];
        const uint64_t hh = (uint64_t)v.y << 32 | v.x;
        if (hh == h) { off = v.z; cnt = v.w; return true; }
        if (hh == HT_EMPTY) return false;
        s = (s + 1) & I.ht_mask;
    }
}

// inclusive prefix sum over the 64 lanes of a wave with DPP moves only (gfx9 pattern: row_shr 1 / 2 / 4 / 8 inside the rows of 16, then row_bcast:15
// into rows 1 and 3 and row_bcast:31 into rows 2 and 3)
__device__ __forceinline__ uint32_t d_wave_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);
    return v;
}

// ---------------------------------------------------------------------------------------
// 3. seeding: one block per query.  MODE 0 counts anchors per minimizer, MODE 1 writes keys.
// Round 6: MODE 0 also turns the counts into offsets RELATIVE TO THE QUERY'S FIRST ANCHOR (a block-level scan over the query's
// minimizers, 256 at a time) and leaves the query's total in q_cnt: what follows is one small scan over the queries (k_qscan)
// instead of the library's scan + 64-bit reduction over all ~24 M minimizers of a range (2.1 ms per range on the stage's critical path).
struct SeedArgs {
    IndexView I;
    const uint64_t *mz_x; const uint32_t *mz_y;
    const int32_t *q_mzoff;      // [nq+1]
    const int32_t *qlen;
    const int32_t *qtarget;      // nullable
    const int32_t *q_order;      // block -> query (longest first), nullable
    int32_t mid_occ;
    const int32_t *tmid;         // per-target occurrence cut-offs [n_targets] (nullable: the pooled cut-off mid_occ applies)
    int32_t per_target;          // qtarget < 0: filter and count a minimizer's occurrences separately inside every target
    int32_t n_targets;
    int32_t *mz_cnt;             // MODE 0 out / MODE 1 in: anchors of the minimizer
    int32_t *mz_ent;             // MODE 0 out / MODE 1 in: first occurrence of the minimizer in `pos` (saves the second probe)
    int32_t *mz_n;               // MODE 0 out / MODE 1 in: its occurrence count (0 = absent)
    int32_t *mz_aoff;            // MODE 0 out / MODE 1 in: its first anchor, counted from the query's first anchor
    int32_t *q_cnt;              // MODE 0 out: anchors of the query
    const int32_t *q_aoff;       // MODE 1 in (keys in memory): first anchor of every query
    uint64_t *keys;              // MODE 1 out
    // staged input (tile_off non-null): mz_x / mz_y are the sketch kernel's per-tile staging arrays (tile t at t * SK_TILE,
    // tile_off[t+1] - tile_off[t] entries), read in place instead of being compacted first; q_tile0[q] = first tile of query q
    const int32_t *tile_off, *q_tile0;
    // MODE 1 inside the LDS sort (segsort.hip.h: SeedProducer): anchor w of the query goes to lds_keys[w] instead of keys[q_aoff[q] + w]
    uint64_t *lds_keys;
};

__device__ __forceinline__ void d_put_key(const SeedArgs &A, int64_t w, uint64_t key)
{
    if (A.lds_keys) A.lds_keys[w] = key; else A.keys[w] = key;
}
// target holding global position g (goff ascending, goff[n] = end)
__device__ __forceinline__ int d_tid_of(const uint32_t *__restrict__ goff, int n, uint32_t g)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (goff[mid] <= g) lo = mid; else hi = mid - 1; }
    return lo;
}

// first index in [lo, hi) of an occurrence list whose global position is >= g (a list is sorted by position: the index sort is stable)
__device__ __forceinline__ uint32_t d_occ_lower(const uint32_t *__restrict__ pos, uint32_t lo, uint32_t hi, uint32_t g)
{
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if ((pos[mid] >> 1) < g) lo = mid + 1; else hi = mid; }
    return lo;
}
// A query restricted to one target (qtarget >= 0) and the per-target mode stand for the reference's separate runs of the
// aligner against every contig (TELR_te.py:68-78,119-132,504-506; TELR_assembly.py:199-212): a minimizer's occurrences are
// counted inside the target and compared with THAT target's cut-off (tmid), so a TE k-mer shared by hundreds of contigs
// is not masked as repetitive.
// the minimizers of query q, taken in turn by `nthr` threads (this one is `tid`)
template <int MODE>
__device__ __forceinline__ void d_seed_query(const SeedArgs &A, const int q, const int tid, const int nthr)
{
    const int m0 = A.q_mzoff[q], m1 = A.q_mzoff[q + 1];
    const int tf = A.qtarget ? A.qtarget[q] : -1;
    const bool pt = tf < 0 && A.per_target && A.tmid;
    uint32_t g0 = 0, g1 = 0xffffffffu;
    if (tf >= 0) { g0 = A.I.goff[tf]; g1 = g0 + (uint32_t)A.I.tlen[tf]; }
    const int32_t occ = (tf >= 0 && A.tmid) ? A.tmid[tf] : A.mid_occ;
    const int qlen = A.qlen[q];
    // staged input: minimizer g of the query lives in tile t at slot g - tile_off[t]; g only grows, so t is advanced, not searched
    int t = A.tile_off ? A.q_tile0[q] : 0, t_lo = 0, t_hi = 0;
    if (A.tile_off) { t_lo = A.tile_off[t]; t_hi = A.tile_off[t + 1]; }
    const int64_t wbase = MODE == 1 && !A.lds_keys ? A.q_aoff[q] : 0;       // where the query's anchors start in the key array
    __shared__ int32_t s_wtot[16];                                           // MODE 0 (one block of at most 1,024 threads per query): the waves' totals
    int32_t carry = 0;                                                       // MODE 0: anchors of the minimizers before this round's
    for (int gb = m0; gb < m1; gb += nthr) {
      int32_t cnt_out = 0;
      const int g = gb + tid;
      if (g < m1) do {
        int64_t gi = g;
        if (A.tile_off) {
            while (g >= t_hi) { ++t; t_lo = t_hi; t_hi = A.tile_off[t + 1]; }
            gi = (int64_t)t * SK_TILE + (g - t_lo);
        }
        uint64_t x = A.mz_x[gi];
        uint32_t o0 = 0, o1 = 0;
        if (MODE == 0) {
            uint32_t off = 0, n = 0;
            if (!d_ht_lookup(A.I, x >> 8, off, n)) n = 0;
            A.mz_ent[g] = (int32_t)off; A.mz_n[g] = (int32_t)n;
            o0 = off; o1 = off + n;
        } else {
            if (A.mz_cnt[g] == 0) break;
            o0 = (uint32_t)A.mz_ent[g]; o1 = o0 + (uint32_t)A.mz_n[g];
        }
        // one occurrence: o0 IS the occurrence (k_ht_build), not an index into `pos`
        const bool single = o1 - o0 == 1u;
        const uint32_t pos1 = o0;
#define SEED_POS(i_) (single ? pos1 : A.I.pos[i_])
        if (pt) {
            // occurrences are sorted by global position, i.e. grouped by target: one run per target
            int32_t total = 0; int64_t w = MODE == 1 ? wbase + A.mz_aoff[g] : 0;
            uint64_t kf = 0, kr = 0; int32_t qz = 0;
            if (MODE == 1) {
                const uint32_t y = A.mz_y[gi];
                const int32_t span = (int32_t)(x & 0xff), qpos = (int32_t)(y >> 1); qz = (int32_t)(y & 1);
                kf = (uint64_t)qpos << 8 | (uint64_t)span;
                kr = (1ULL << 63) | (uint64_t)(qlen - (qpos + 1 - span) - 1) << 8 | (uint64_t)span;
            }
            uint32_t o = o0;
            while (o < o1) {
                const int t = d_tid_of(A.I.goff, A.n_targets, SEED_POS(o) >> 1);
                const uint32_t gend = A.I.goff[t + 1];
                uint32_t e = o + 1;
                // (a run is one to three occurrences as a rule; a satellite's thousands inside one contig are stepped over by bisection)
                for (int lin = 0; e < o1 && (SEED_POS(e) >> 1) < gend; ) { ++e; if (++lin == 8) { e = d_occ_lower(A.I.pos, e, o1, gend); break; } }
                if ((int32_t)(e - o) <= A.tmid[t]) {
                    total += (int32_t)(e - o);
                    if (MODE == 1) for (uint32_t z = o; z < e; ++z) { const uint32_t py = SEED_POS(z); d_put_key(A, w++, ((int)(py & 1) == qz ? kf : kr) | (uint64_t)(py >> 1) << 32); }
                }
                o = e;
            }
            cnt_out = total;
            break;
        }
        int32_t cnt = 0;
        uint32_t a0 = o0, a1 = o1;           // the occurrences that count: all, or those inside the query's target
        if (o1 > o0) {
            if (tf >= 0) {
                // Round 6: the target's piece of the list by bisection (the list is sorted by position).  A linear sweep met lists of
                // 10^5-10^6 entries on the hard genome -- a satellite shared by the contigs of 150 loci, S6 -- once per query minimizer.
                if (single) { const uint32_t gp = pos1 >> 1; if (gp < g0 || gp >= g1) a1 = a0; }
                else { a0 = d_occ_lower(A.I.pos, o0, o1, g0); a1 = d_occ_lower(A.I.pos, a0, o1, g1); }
            }
            cnt = (int32_t)(a1 - a0);
            if (cnt > occ) cnt = 0;
        }
        cnt_out = cnt;
        if (MODE == 1 && cnt > 0) {
            uint32_t y = A.mz_y[gi];
            int32_t span = (int32_t)(x & 0xff), qpos = (int32_t)(y >> 1), qz = (int32_t)(y & 1);
            uint64_t kf = (uint64_t)qpos << 8 | (uint64_t)span;
            uint64_t kr = (1ULL << 63) | (uint64_t)(qlen - (qpos + 1 - span) - 1) << 8 | (uint64_t)span;
            int64_t w = wbase + A.mz_aoff[g];
            for (uint32_t o = a0; o < a1; ++o) {
                const uint32_t py = SEED_POS(o);
                d_put_key(A, w++, ((int)(py & 1) == qz ? kf : kr) | (uint64_t)(py >> 1) << 32);
            }
        }
      } while (0);
      if (MODE == 0 && g < m1) A.mz_cnt[g] = cnt_out;
    }
    if (MODE == 0) {
        // the counts -> offsets from the query's first anchor, in a pass of its own behind the probes (with the scan inside the probing
        // rounds every round ended in a barrier and the block's waves could no longer hide each other's table misses: 8.1 -> 9.9 ms
        // per range): 256 counts a round, wave scan (DPP), the waves' totals through LDS
        __syncthreads();
        const int wv = tid >> 6, nwv = (nthr + 63) >> 6;
        for (int gb = m0; gb < m1; gb += nthr) {
            const int g = gb + tid;
            const int32_t c = g < m1 ? A.mz_cnt[g] : 0;
            const uint32_t inc = d_wave_scan_add((uint32_t)c);
            if ((tid & 63) == 63) s_wtot[wv] = (int32_t)inc;
            __syncthreads();
            int32_t before = 0, all = 0;
            for (int z = 0; z < nwv; ++z) { const int32_t v = s_wtot[z]; if (z < wv) before += v; all += v; }
            if (g < m1) A.mz_aoff[g] = carry + before + (int32_t)inc - c;
            carry += all;
            __syncthreads();
        }
        if (tid == 0) A.q_cnt[q] = carry;
    }
}
#undef SEED_POS
template <int MODE>
__global__ void __launch_bounds__(256) k_seed(SeedArgs A)
{
    d_seed_query<MODE>(A, A.q_order ? A.q_order[blockIdx.x] : blockIdx.x, threadIdx.x, blockDim.x);
}

// ---- seeding with sub-read voting (spec 3.10: NGMLR's candidate search) ------------------------------------------------
// All-vs-all calls with mo.vote_len > 0, plain (compacted) minimizer arrays.
//   k_vote_lookup   one thread per minimizer: table probe, occurrence cut-off -> (first occurrence, count); flat and wide, so
//                   the scattered probes have thousands of waves to hide behind
//   k_vote_qhits    hits per query (an upper bound of its anchors) -> after a scan, the query's piece of the staging array
//   k_seed_vote     one block (three waves) per query; the waves take its sub-reads in turn (minimizers are in query order: a
//                   sub-read is a contiguous run of them, found by one boundary sweep into LDS).  A wave lays the hit lists
//                   of up to 128 minimizers end to end and walks them ONE LANE PER HIT (a lane finds its list by counting the
//                   list starts up to its hit: bits of one word per 64-hit window, set by the lanes that own the lists; two
//                   windows are in flight so that their occurrence loads overlap):
//                   every hit votes for its folded diagonal bin in the wave's LDS table (one atomic; the fullest bin comes
//                   back with the atomics' returns) and is remembered in LDS.  Then the remembered hits are tested -- a hit
//                   stays iff its bin and the two neighbours hold enough votes -- and the survivors go to the query's piece
//                   of the staging array (block-level cursor; their order is irrelevant, the keys are sorted next).
//                   Sub-reads with more hits than the LDS record holds walk the lists a second time instead.
//   k_vote_compact  staging -> the dense key arrays at the scanned per-query offsets
#define VOTE_SLOTS   2048
#define VOTE_SUBCAP  512
#ifndef VOTE_WAVES
#define VOTE_WAVES   3
#endif
#ifndef VOTE_HCAP
#define VOTE_HCAP    768            /* hits of a sub-read remembered in LDS */
#endif
#define VOTE_MZ      128            /* minimizers per chunk (two per lane) */
struct VoteOpt { int32_t len, shift, vmin, frac_q8; };
struct VoteArgs {
    const int64_t *q_soff;       // [nq+1] start of every query's piece of the staging array (scan of the hit counts)
    uint64_t *stage;             // survivors
    int32_t *q_cnt;              // out: survivors per query
};
__device__ __forceinline__ uint32_t d_vote_slot(uint32_t gp, uint32_t qadj, uint32_t rev, int shift)
{
    const uint32_t d = gp - qadj + (1u << 24);
    return (((d >> shift) & (VOTE_SLOTS / 2 - 1)) << 1) | rev;
}
// FILTER = false: short k-mers (the ngmlr-* presets' 13-mers: 67 M possible, most of them in a 100-Mb genome) -- the filter
// bitmap answers "maybe" for nearly every probe and only costs its own scattered line
template <bool FILTER>
__global__ void __launch_bounds__(256) k_vote_lookup(IndexView I, const uint64_t *__restrict__ mz_x, int32_t nmz, int32_t mid_occ, int32_t *__restrict__ mz_ent, int32_t *__restrict__ mz_n)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= nmz) return;
    uint32_t off = 0, n = 0;
    if (!d_ht_lookup<FILTER>(I, mz_x[g] >> 8, off, n)) n = 0;
    if (n > (uint32_t)mid_occ) n = 0;
    mz_ent[g] = (int32_t)off; mz_n[g] = (int32_t)n;
}
__global__ void __launch_bounds__(64) k_vote_qhits(const int32_t *__restrict__ q_mzoff, const int32_t *__restrict__ mz_n, int32_t nq, int64_t *__restrict__ q_hits)
{
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q > nq) return;
    int64_t s = 0;
    if (q < nq) for (int g = q_mzoff[q] + lane; g < q_mzoff[q + 1]; g += 64) s += mz_n[g];
    for (int o = 32; o >= 1; o >>= 1) s += (int64_t)((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)s, o) | (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)((uint64_t)s >> 32), o) << 32);
    if (lane == 0) q_hits[q] = s;
}
#ifndef VOTE_WIN
#define VOTE_WIN 8                 /* windows of 64 hits whose occurrence loads are in flight together */
#endif
struct VoteChunk { uint32_t P[VOTE_MZ], off[VOTE_MZ], qpos[VOTE_MZ]; uint16_t zs[VOTE_MZ]; uint8_t rid[VOTE_MZ]; uint32_t wm[2 * VOTE_WIN]; };      // rid[r] = r-th minimizer WITH hits; wm: list-start bits of the two hit windows in flight
struct VoteHit { uint32_t gp; uint32_t sm; };      // sm = slot | minimizer << 11
// The vote table of a wave: 2,048 counters.  T16: two 16-bit counters per word -- half the LDS, 12 instead of 9 waves per CU -- for
// queries whose hits cannot overflow one (the kernel is launched twice: queries with at most `lim16` hits here, the others with
// 32-bit counters; slot s and slot s ^ 1, the two strands of a diagonal bin, share a word).
template <bool T16> __device__ __forceinline__ uint32_t d_vt_add(uint32_t *T, uint32_t s)
{
    if (T16) { const uint32_t sh = (s & 1u) << 4; return ((atomicAdd(&T[s >> 1], 1u << sh) >> sh) & 0xffffu) + 1u; }
    return atomicAdd(&T[s], 1u) + 1u;
}
template <bool T16> __device__ __forceinline__ void d_vt_vote(uint32_t *T, uint32_t s)
{
    if (T16) (void)__hip_atomic_fetch_add(&T[s >> 1], 1u << ((s & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else (void)__hip_atomic_fetch_add(&T[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <bool T16> __device__ __forceinline__ uint32_t d_vt_get(const uint32_t *T, uint32_t s)
{
    return T16 ? (T[s >> 1] >> ((s & 1u) << 4)) & 0xffffu : T[s];
}
template <bool T16>
__global__ void __launch_bounds__(64 * VOTE_WAVES) k_seed_vote(SeedArgs A, VoteOpt V, VoteArgs VA, int64_t lim16)
{
    constexpr int TW = T16 ? VOTE_SLOTS / 2 : VOTE_SLOTS;          // words of a wave's table
    {   // this launch's share of the queries
        const int q_ = A.q_order ? A.q_order[blockIdx.x] : blockIdx.x;
        const bool small = VA.q_soff[q_ + 1] - VA.q_soff[q_] <= lim16;
        if (small != T16) return;
    }
    __shared__ uint32_t tab[VOTE_WAVES][TW];
    __shared__ VoteHit stash[VOTE_WAVES][VOTE_HCAP];
    __shared__ VoteChunk chunk[VOTE_WAVES];
    __shared__ int32_t sub_first[VOTE_SUBCAP + 1];
    __shared__ uint32_t blk_cnt;
    const int q = A.q_order ? A.q_order[blockIdx.x] : blockIdx.x;
    const int m0 = A.q_mzoff[q], m1 = A.q_mzoff[q + 1];
    const int qlen = A.qlen[q];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // (a scalar: the sub-read loop, its chunk bounds and the wave's LDS bases stay on the scalar side)
    uint32_t *T = tab[wv];
    VoteHit *HS = stash[wv];
    VoteChunk &C = chunk[wv];
    uint64_t *out = VA.stage + VA.q_soff[q];
    for (int i = lane; i < TW; i += 64) T[i] = 0;
    if (tid == 0) blk_cnt = 0;
    const int nsub = (qlen + V.len - 1) / V.len;
    for (int sb = 0; sb < nsub; sb += VOTE_SUBCAP) {
        // first minimizer of every sub-read sb .. sb + VOTE_SUBCAP (sub_first[j] = first g whose sub-read is >= sb + j)
        __syncthreads();
        for (int i = tid; i <= VOTE_SUBCAP; i += 64 * VOTE_WAVES) sub_first[i] = m1;
        __syncthreads();
        for (int g = m0 + tid; g < m1; g += 64 * VOTE_WAVES) {
            const int s = (int)(A.mz_y[g] >> 1) / V.len;
            const int sp = g > m0 ? (int)(A.mz_y[g - 1] >> 1) / V.len : -1;
            if (s != sp) {
                int lo = sp + 1 > sb ? sp + 1 : sb, hi = s < sb + VOTE_SUBCAP ? s : sb + VOTE_SUBCAP;
                for (int j = lo; j <= hi; ++j) sub_first[j - sb] = g;
            }
        }
        __syncthreads();
        const int send = nsub < sb + VOTE_SUBCAP ? nsub : sb + VOTE_SUBCAP;
        // (the first chunk of the wave's NEXT sub-read is loaded while the current one is voted: its round trip is off the wave's path)
        uint32_t pf_off[2] = {0, 0}, pf_n[2] = {0, 0}, pf_y[2] = {0, 0}, pf_sp[2] = {0, 0}; int pf_s = -1;
        // VOTE_LOAD_CHUNK only ISSUES the loads (masked loads into zeroed registers: nothing depends on them here); VOTE_PIN, at the
        // place of use, keeps the compiler from carrying derived values instead -- it would compute them, and wait, right behind the loads
#define VOTE_LOAD_CHUNK(o_, n_, y_, sp_, gc_, g1_) { _Pragma("unroll") for (int u = 0; u < 2; ++u) { const int g = (gc_) + 2 * lane + u; o_[u] = 0; n_[u] = 0; y_[u] = 0; sp_[u] = 0; \
            if (g < (g1_)) { o_[u] = (uint32_t)A.mz_ent[g]; n_[u] = (uint32_t)A.mz_n[g]; y_[u] = A.mz_y[g]; sp_[u] = ((const uint32_t*)A.mz_x)[2 * (size_t)g]; } } }
#define VOTE_PIN(o_, n_, y_, sp_) { _Pragma("unroll") for (int u = 0; u < 2; ++u) asm volatile("" : "+v"(o_[u]), "+v"(n_[u]), "+v"(y_[u]), "+v"(sp_[u])); }
        for (int s = sb + wv; s < send; s += VOTE_WAVES) {
            const int g0 = __builtin_amdgcn_readfirstlane(sub_first[s - sb]), g1 = __builtin_amdgcn_readfirstlane(sub_first[s - sb + 1]);
            uint32_t c0_off[2], c0_n[2], c0_y[2], c0_sp[2];
            if (pf_s == s) {
                VOTE_PIN(pf_off, pf_n, pf_y, pf_sp)
#pragma unroll
                for (int u = 0; u < 2; ++u) { c0_off[u] = pf_off[u]; c0_n[u] = pf_n[u]; c0_y[u] = pf_y[u]; c0_sp[u] = pf_sp[u]; }
            } else VOTE_LOAD_CHUNK(c0_off, c0_n, c0_y, c0_sp, g0, g1)
            if (s + VOTE_WAVES < send) { pf_s = s + VOTE_WAVES; const int pg0 = __builtin_amdgcn_readfirstlane(sub_first[pf_s - sb]), pg1 = __builtin_amdgcn_readfirstlane(sub_first[pf_s - sb + 1]); VOTE_LOAD_CHUNK(pf_off, pf_n, pf_y, pf_sp, pg0, pg1) }
            if (g0 >= g1) continue;
            // ---- votes (all chunks of the sub-read), hits remembered
            uint32_t vmax = 0, nhit = 0;
            for (int pass = 0; pass < 2; ++pass) {
                // pass 0: vote.  pass 1: only when the sub-read has more hits than the LDS record -- walk again and test against the table
                if (pass == 1 && nhit <= VOTE_HCAP) break;
                uint32_t thr1 = 0;
                if (pass == 1) { thr1 = (vmax * (uint32_t)V.frac_q8 + 255u) >> 8; if (thr1 < (uint32_t)V.vmin) thr1 = (uint32_t)V.vmin; }
                uint32_t hbase = 0;
                for (int gc = g0; gc < g1; gc += VOTE_MZ) {
                    // the chunk: two minimizers per lane
                    uint32_t n2[2];
                    __builtin_amdgcn_wave_barrier();
                    uint32_t c_off[2], c_y[2], c_sp[2];
                    if (gc == g0) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) { c_off[u] = c0_off[u]; n2[u] = c0_n[u]; c_y[u] = c0_y[u]; c_sp[u] = c0_sp[u]; }
                    } else VOTE_LOAD_CHUNK(c_off, n2, c_y, c_sp, gc, g1)
#pragma unroll
                    for (int u = 0; u < 2; ++u) { C.qpos[2 * lane + u] = c_y[u] >> 1; C.zs[2 * lane + u] = (uint16_t)((c_y[u] & 1u) << 8 | (c_sp[u] & 0xffu)); }
                    uint32_t inc = n2[0] + n2[1];
                    inc = d_wave_scan_add(inc);          // (DPP row shifts + the two row broadcasts: no LDS round trips; six ds_bpermute before)
                    // the minimizers that have hits, in order (rid), and where this lane's two lists start: a hit's list is then found
                    // by counting list starts up to it -- bits of one 64-bit word per window -- instead of bisecting the prefix sums.
                    // Round 6: the list records are indexed by that RANK (P[r] = the hits up to and including list r, off[r] = its first
                    // occurrence), so a hit reads its list's bounds, its occurrence offset and its minimizer side by side -- ONE level of LDS
                    // reads in front of the occurrence load, where the minimizer's number stood between them (round 5's list, item 2)
                    const uint64_t nz0 = __ballot(n2[0] > 0), nz1 = __ballot(n2[1] > 0);
                    const uint64_t below = (1ULL << lane) - 1ULL;
                    const uint32_t rk0 = (uint32_t)__popcll(nz0 & below) + (uint32_t)__popcll(nz1 & below);
                    if (n2[0]) { C.rid[rk0] = (uint8_t)(2 * lane); C.P[rk0] = inc - n2[1]; C.off[rk0] = c_off[0]; }
                    if (n2[1]) { const uint32_t rk1 = rk0 + (n2[0] ? 1u : 0u); C.rid[rk1] = (uint8_t)(2 * lane + 1); C.P[rk1] = inc; C.off[rk1] = c_off[1]; }
                    const uint32_t st0 = inc - n2[0] - n2[1], st1 = inc - n2[1];
                    uint32_t cb = 0;              // lists that start before the current window
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t H = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                    // the walk: VOTE_WIN windows of 64 hits per trip.  Round 5: EIGHT windows (four before): with in-kernel clocks a trip of two
                    // windows took 3,700 cycles -- one scattered-load latency of the loaded part: a trip decodes its hits' lists, ASKS for their
                    // occurrences and waits -- and a sub-read (430 hits on average) made four of them in a row; now a sub-read's occurrence
                    // loads go out together (SQ_WAIT_ANY was 60 % of the kernel's wave cycles) and are consumed window by window as they arrive.
                    for (uint32_t h0 = 0; h0 < H; h0 += 64 * VOTE_WIN) {
                        uint32_t m_[VOTE_WIN], py_[VOTE_WIN]; bool in_[VOTE_WIN];
                        // list starts inside the windows of this trip
                        if (lane < 2 * VOTE_WIN) C.wm[lane] = 0;
                        __builtin_amdgcn_wave_barrier();
                        { const uint32_t d0 = st0 - h0, d1 = st1 - h0;
                          if (n2[0] && d0 < 64u * VOTE_WIN) atomicOr(&C.wm[d0 >> 5], 1u << (d0 & 31));
                          if (n2[1] && d1 < 64u * VOTE_WIN) atomicOr(&C.wm[d1 >> 5], 1u << (d1 & 31)); }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int u = 0; u < VOTE_WIN; ++u) {
                            const uint32_t h = h0 + 64 * u + lane;
                            in_[u] = h < H; m_[u] = 0; py_[u] = 0;
                            if (h0 + 64 * u < H) {          // (wave-uniform: a window past the last hit asks for nothing)
                                const uint64_t wmask = (uint64_t)C.wm[2 * u + 1] << 32 | C.wm[2 * u];
                                const uint32_t upto = (uint32_t)__popcll(wmask & ((2ULL << lane) - 1ULL));
                                if (in_[u]) {
                                    const uint32_t r = cb + upto - 1u;
                                    const uint32_t pm = r ? C.P[r - 1] : 0u, nm = C.P[r] - pm, of = C.off[r];
                                    m_[u] = C.rid[r];
                                    py_[u] = nm == 1 ? of : A.I.pos[of + (h - pm)];
                                }
                                cb += (uint32_t)__popcll(wmask);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < VOTE_WIN; ++u) {
                            if (h0 + 64 * u >= H) break;
                            const uint32_t h = h0 + 64 * u + lane;
                            bool pass_hit = false; uint32_t gp_ = 0, rv_ = 0;
                            if (in_[u]) {
                                const uint32_t m = m_[u], zs = C.zs[m], span = zs & 0xffu, qpos = C.qpos[m];
                                rv_ = (py_[u] & 1u) ^ (zs >> 8); gp_ = py_[u] >> 1;
                                const uint32_t qadj = rv_ ? (uint32_t)qlen - (qpos + 1 - span) - 1 : qpos;
                                const uint32_t sl_ = d_vote_slot(gp_, qadj, rv_, V.shift);
                                if (pass == 0) {
                                    d_vt_vote<T16>(T, sl_);          // (no return value: the fullest bin is read off the table afterwards)
                                    if (hbase + h < VOTE_HCAP) { VoteHit x; x.gp = gp_; x.sm = sl_ | m << 11 | (uint32_t)(gc - g0) / VOTE_MZ << 18; HS[hbase + h] = x; }
                                } else pass_hit = d_vt_get<T16>(T, (sl_ - 2u) & (VOTE_SLOTS - 1)) + d_vt_get<T16>(T, sl_) + d_vt_get<T16>(T, (sl_ + 2u) & (VOTE_SLOTS - 1)) >= thr1;
                            }
                            if (pass == 1) {
                                const uint64_t bm = __ballot(pass_hit);
                                uint32_t base = 0;
                                if (lane == 0 && bm) base = atomicAdd(&blk_cnt, (uint32_t)__popcll(bm));
                                base = (uint32_t)__shfl((int)base, 0);
                                if (pass_hit) {
                                    const uint32_t m = m_[u], zs = C.zs[m], span = zs & 0xffu, qpos = C.qpos[m];
                                    const uint64_t key = (rv_ ? (1ULL << 63) | (uint64_t)((uint32_t)qlen - (qpos + 1 - span) - 1) << 8 : (uint64_t)qpos << 8) | (uint64_t)span | (uint64_t)gp_ << 32;
                                    out[base + (uint32_t)__popcll(bm & ((1ULL << lane) - 1ULL))] = key;
                                }
                            }
                        }
                    }
                    hbase += H;
                }
                if (pass == 0) {
                    nhit = hbase;
                    if (nhit > VOTE_HCAP) {          // (rare: the second walk needs the fullest bin first -- one sweep over the table)
                        __builtin_amdgcn_wave_barrier();
                        for (int i = lane; i < TW; i += 64) { const uint32_t w_ = T[i]; const uint32_t a_ = T16 ? (w_ & 0xffffu) : w_, b_ = T16 ? (w_ >> 16) : 0u; vmax = a_ > vmax ? a_ : vmax; vmax = b_ > vmax ? b_ : vmax; }
                        for (int o = 32; o >= 1; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)vmax, o); vmax = v > vmax ? v : vmax; }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            const bool one_chunk = g1 - g0 <= VOTE_MZ;
            if (nhit <= VOTE_HCAP) {
                // ---- survivors from the remembered hits.  The minimizer data of a remembered hit is read from the mz arrays again
                // (the LDS chunk holds only the last 128 minimizers): qpos / span / strand by minimizer index g0 + chunk * 128 + m
                // Round 5: the votes are fire-and-forget additions (no returned count to wait for in the walk); here the groups of 64 remembered
                // hits are read TOGETHER -- the hit, its bin and the two neighbours: independent LDS reads, back to back -- the fullest bin
                // is the largest centre count any hit sees (every voted bin is some hit's bin), the survivors of the whole sub-read take
                // their place in the query's piece with ONE addition to the block's cursor, then the keys are written.
                constexpr int NG = VOTE_HCAP / 64;
                VoteHit xs[NG]; uint32_t s3[NG]; uint32_t cmax = 0;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    xs[g].gp = 0; xs[g].sm = 0; s3[g] = 0;
                    if ((uint32_t)(g * 64) < nhit) {
                        const uint32_t h = (uint32_t)(g * 64) + lane;
                        if (h < nhit) {
                            xs[g] = HS[h];
                            const uint32_t sl = xs[g].sm & (VOTE_SLOTS - 1);
                            const uint32_t c_ = d_vt_get<T16>(T, sl);
                            s3[g] = d_vt_get<T16>(T, (sl - 2u) & (VOTE_SLOTS - 1)) + c_ + d_vt_get<T16>(T, (sl + 2u) & (VOTE_SLOTS - 1));
                            cmax = c_ > cmax ? c_ : cmax;
                        }
                    }
                }
                for (int o = 32; o >= 1; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)cmax, o); cmax = v > cmax ? v : cmax; }
                vmax = cmax;
                uint32_t thr = (vmax * (uint32_t)V.frac_q8 + 255u) >> 8;
                if (thr < (uint32_t)V.vmin) thr = (uint32_t)V.vmin;
                uint64_t bm[NG]; uint32_t tot = 0;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    bm[g] = 0;
                    if ((uint32_t)(g * 64) < nhit) { bm[g] = __ballot((uint32_t)(g * 64) + lane < nhit && s3[g] >= thr); tot += (uint32_t)__popcll(bm[g]); }
                }
                uint32_t base = 0;
                if (lane == 0 && tot) base = atomicAdd(&blk_cnt, tot);
                base = (uint32_t)__shfl((int)base, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if ((uint32_t)(g * 64) < nhit) {
                        if ((bm[g] >> lane) & 1ULL) {
                            // qpos / span of the hit's minimizer: still in the LDS chunk when the sub-read is ONE chunk (nearly always:
                            // 256 bases hold ~85 (w,k) = (5,13) minimizers), else from the arrays again
                            const VoteHit x = xs[g];
                            const uint32_t mi = (x.sm >> 11) & (VOTE_MZ - 1);
                            uint32_t span, qpos;
                            if (one_chunk) { span = C.zs[mi] & 0xffu; qpos = C.qpos[mi]; }
                            else { const int gq = g0 + (int)(x.sm >> 18) * VOTE_MZ + (int)mi; span = (uint32_t)(A.mz_x[gq] & 0xff); qpos = A.mz_y[gq] >> 1; }
                            const uint32_t rev = x.sm & 1u;
                            const uint64_t key = (rev ? (1ULL << 63) | (uint64_t)((uint32_t)qlen - (qpos + 1 - span) - 1) << 8 : (uint64_t)qpos << 8) | (uint64_t)span | (uint64_t)x.gp << 32;
                            out[base + (uint32_t)__popcll(bm[g] & ((1ULL << lane) - 1ULL))] = key;
                        }
                        base += (uint32_t)__popcll(bm[g]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                for (uint32_t h = lane; h < nhit; h += 64) { const uint32_t sl = HS[h].sm & (VOTE_SLOTS - 1); T[T16 ? sl >> 1 : sl] = 0; }
            } else for (int i = lane; i < TW; i += 64) T[i] = 0;
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    if (tid == 0) VA.q_cnt[q] = (int32_t)blk_cnt;
}
#undef VOTE_LOAD_CHUNK
#undef VOTE_PIN
// staging -> dense keys (only ahead of the library sort: the LDS sort of segsort.hip.h reads the staging pieces in place)
__global__ void __launch_bounds__(256) k_vote_compact(const uint64_t *__restrict__ stage, const int64_t *__restrict__ q_soff, const int32_t *__restrict__ q_aoff, int32_t nq,
                                                      uint64_t *__restrict__ keys, const int32_t *__restrict__ list)
{
    if ((int)blockIdx.x >= nq) return;
    const int q = list ? list[blockIdx.x] : (int)blockIdx.x;        // list: only these queries (the over-size ones ahead of the library sort)
    const int64_t s0 = q_soff[q]; const int a0 = q_aoff[q], n = q_aoff[q + 1] - a0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const uint64_t k = stage[s0 + i];
        keys[a0 + i] = k;
    }
}

// ---- per-target occurrence counts of an index (for the per-target cut-offs) ----------------------------
// run = the occurrences of one minimizer inside one target (contiguous in `pos`: sorted by hash, then position)
__global__ void k_pt_entry_heads(const uint32_t *__restrict__ ent_off, int32_t n_ent, int32_t *__restrict__ head)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_ent) head[ent_off[e]] = 1;
}
__global__ void k_pt_run_heads(const uint32_t *__restrict__ pos, int64_t n, const uint32_t *__restrict__ goff, int n_targets, int32_t *__restrict__ head, int32_t *__restrict__ tid)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = d_tid_of(goff, n_targets, pos[i] >> 1);
    tid[i] = t;
    if (i > 0 && !head[i] && d_tid_of(goff, n_targets, pos[i - 1] >> 1) != t) head[i] = 1;
}
__global__ void k_pt_run_starts(const int32_t *__restrict__ head, const int32_t *__restrict__ rank, int64_t n, int32_t *__restrict__ start, int32_t n_runs)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && head[i]) start[rank[i]] = (int32_t)i;
    if (i == 0) start[n_runs] = (int32_t)n;
}
__global__ void k_pt_run_keys(const int32_t *__restrict__ start, const int32_t *__restrict__ tid, int32_t n_runs, uint64_t *__restrict__ key)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_runs) key[r] = (uint64_t)(uint32_t)tid[start[r]] << 32 | (uint32_t)(start[r + 1] - start[r]);
}
// first run of every target in the sorted keys
__global__ void k_pt_target_off(const uint64_t *__restrict__ key, int32_t n_runs, int32_t n_targets, int32_t *__restrict__ off)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_targets) return;
    const uint64_t want = (uint64_t)(uint32_t)t << 32;
    int lo = 0, hi = n_runs;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (key[mid] < want) lo = mid + 1; else hi = mid; }
    off[t] = lo;
}
// the -f quantile of a target's own counts, +1, clamped: the same rule as the pooled cut-off (index_mid_occ)
__global__ void k_pt_mid_occ(const uint64_t *__restrict__ key, const int32_t *__restrict__ off, int32_t n_targets, float frac, int32_t lo_occ, int32_t hi_occ, int32_t *__restrict__ tmid)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_targets) return;
    const int64_t n = off[t + 1] - off[t];
    int32_t occ;
    if (n == 0) occ = lo_occ;
    else {
        int64_t idx = (int64_t)((1.0 - (double)frac) * (double)n);
        if (idx >= n) idx = n - 1;
        occ = (int32_t)(uint32_t)(key[off[t] + idx] & 0xffffffffu) + 1;
    }
    if (occ < lo_occ) occ = lo_occ;
    if (hi_occ > lo_occ && occ > hi_occ) occ = hi_occ;
    tmid[t] = occ;
}

__global__ void k_gather_i32(const int32_t *__restrict__ src, const int32_t *__restrict__ idx, int32_t n, int32_t tail, int32_t *__restrict__ dst)
{
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
    if (i == n) dst[n] = tail;
}

// ---- exclusive scan of per-query counts (round 6: the per-query offsets of a range -- 10^5 queries -- do not need the library's
// device-wide scan, its state initialisation and a 64-bit reduction beside it).  Two launches of 256-thread workgroups over tiles of
// 4,096 counts: k_qscan_sums (every tile's total, 64-bit; counts above `cap` appended to over_list), k_qscan_write (a tile's base =
// the totals of the tiles before it, then its counts in place).  out[0 .. n] (out[n] = the total, as Tout), tot64 = the total in 64
// bits (an int32 scan wraps silently: the caller halves a batch of 2^31 anchors or more), n_over = counts above `cap` (queries whose
// anchors one workgroup cannot sort in LDS; the tiles add to it: the caller clears it), over_list = their indices in no particular order.
// (A first form, ONE workgroup of 1,024 threads, cost the configs[2] step 11 ms: while the other range's k_dp_pk holds 452 of a SIMD's
// 512 registers no CU has room for sixteen more waves at once, and the little kernel waited milliseconds for a CU to drain.)
#define QSCAN_TILE 4096
template <typename Tin>
__global__ void __launch_bounds__(256) k_qscan_sums(const Tin *__restrict__ cnt, int32_t n, int64_t cap, int64_t *__restrict__ tile_sum, int32_t *__restrict__ n_over, int32_t *__restrict__ over_list)
{
    __shared__ int64_t s[4];
    const int tid = threadIdx.x, base = blockIdx.x * QSCAN_TILE;
    int64_t sum = 0;
#pragma unroll 4
    for (int j = 0; j < QSCAN_TILE / 256; ++j) {
        const int i = base + j * 256 + tid;
        if (i < n) { const int64_t v = (int64_t)cnt[i]; sum += v; if (n_over && v > cap) { const int z = atomicAdd(n_over, 1); if (over_list) over_list[z] = i; } }
    }
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    if ((tid & 63) == 0) s[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) tile_sum[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <typename Tin, typename Tout>
__global__ void __launch_bounds__(256) k_qscan_write(const Tin *__restrict__ cnt, int32_t n, const int64_t *__restrict__ tile_sum, Tout *__restrict__ out, int64_t *__restrict__ tot64)
{
    __shared__ int64_t s[256];
    const int tid = threadIdx.x, base = blockIdx.x * QSCAN_TILE, ntile = gridDim.x;
    // the tiles before this one (and, for the last tile, all of them: the total)
    int64_t bsum = 0, all = 0;
    for (int t = tid; t < ntile; t += 256) { const int64_t v = tile_sum[t]; all += v; if (t < (int)blockIdx.x) bsum += v; }
    s[tid] = bsum;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) { if (tid < o) s[tid] += s[tid + o]; __syncthreads(); }
    const int64_t tile_base = s[0];
    __syncthreads();
    if ((int)blockIdx.x == ntile - 1) {
        s[tid] = all;
        __syncthreads();
        for (int o = 128; o >= 1; o >>= 1) { if (tid < o) s[tid] += s[tid + o]; __syncthreads(); }
        if (tid == 0) { out[n] = (Tout)s[0]; if (tot64) *tot64 = s[0]; }
        __syncthreads();
    }
    // thread t owns 16 consecutive counts of the tile
    const int lo = base + tid * (QSCAN_TILE / 256);
    int64_t v[QSCAN_TILE / 256], sum = 0;
#pragma unroll
    for (int j = 0; j < QSCAN_TILE / 256; ++j) { v[j] = lo + j < n ? (int64_t)cnt[lo + j] : 0; sum += v[j]; }
    s[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int64_t x = tid >= d ? s[tid - d] : 0;
        __syncthreads();
        s[tid] += x;
        __syncthreads();
    }
    int64_t run = tile_base + s[tid] - sum;
#pragma unroll
    for (int j = 0; j < QSCAN_TILE / 256; ++j) { if (lo + j < n) out[lo + j] = (Tout)run; run += v[j]; }
}

// ---------------------------------------------------------------------------------------
// 4. chaining.  f(i) = max(span_i, max_{i-H<=j<i} f(j) + sc(j,i)), largest j among ties.
//    One wave per query, "forward push": lane l owns the anchors i == l (mod 64) of the
//    next R*64 indices in registers; at step j the owner lane publishes (key_j, f_j) by a
//    lane read and every lane scores its own anchors against j.  No LDS, no reduction.
#define A_G(k)    ((int32_t)(((k) >> 32) & 0x7fffffff))
#define A_Q(k)    ((int32_t)(((k) >> 8) & 0xffffff))
#define A_SPAN(k) ((int32_t)((k) & 0xff))

__device__ __forceinline__ uint64_t d_readlane64(uint64_t v, int lane)
{
    uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return (uint64_t)hi << 32 | lo;
}

// One link in the chaining loop: own anchor (gi, qi, span-1) <- broadcast anchor j (gj+1, qj+1, 2 f_j + 2).  The oracle's
// chain_sc() plus the max/tie rule in ~19 instructions:
//  * the strand sits in bit 31 of the reference word, so a strand mismatch fails the unsigned range test of dr;
//  * dr-1 and dq-1 come straight from the subtraction (the +1 is added to the broadcast scalars), so 0 < d <= max_gap is
//    one unsigned compare d-1 < max_gap, and "dd <= bw" joins the same compare as dd + (max_gap-1-bw) (it is implied when
//    bw >= max_gap): ok = max3(dr-1, dq-1, dd + ddc) < max_gap;  |dr-dq| is one v_sad_u32;  min(span, dr, dq) one v_min3;
//  * the Q8 integer log2 of dd+1 is read off the float conversion (exact below 2^24): bits >> 15 = (127 + e) << 8 | the top
//    8 mantissa bits, so ilog2_q8 >> 1 = (bits >> 16) - 16256; the products fit 24 bits whenever the link is allowed;
//  * the running maximum is kept as B = 2 best + (1 while no predecessor was taken), which turns "larger, or equal and the
//    current best already has a predecessor" (ascending j, the largest j wins ties) into the single compare 2 v >= B.
// With chain_skip_q8 == 0 (every preset) the oracle's "dd || dg > span" condition is moot (the penalty is 0 when dd == 0).
template <bool SKIP>
__device__ __forceinline__ void d_chain_push(uint32_t gi, uint32_t qi, uint32_t sp1, uint32_t gj1, uint32_t qj1, int32_t fj2p2, int32_t j,
                                             uint32_t max_gap, uint32_t ddc, uint32_t gap_q8, uint32_t skip_q8, int32_t &B, int32_t &bp)
{
    const uint32_t dr1 = gi - gj1, dq1 = qi - qj1;
    uint32_t dd;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(dd) : "v"(dr1), "v"(dq1));
    const uint32_t dgm = dr1 < dq1 ? dr1 : dq1;
    const uint32_t vm = sp1 < dgm ? sp1 : dgm;                                  // min(span, dg) - 1
    uint32_t t = dr1 > dq1 ? dr1 : dq1; const uint32_t u = dd + ddc; t = t > u ? t : u;
    const bool ok = t < max_gap;
    const uint32_t l2h = __float_as_uint((float)(dd + 1u)) >> 16;              // ilog2_q8(dd+1)/2 + 16256
    int32_t pen = (int32_t)(__umul24(gap_q8, dd) + l2h) - 16256;
    if (SKIP) {
        pen += (int32_t)__umul24(skip_q8, dgm + 1u);
        if (dd == 0u && dgm <= sp1) pen = 0;
    }
    const int32_t v2 = (((int32_t)vm - (pen >> 8)) << 1) + fj2p2;
    if (ok && v2 >= B) { B = v2; bp = j; }
}

// one run of anchors [base, base + n): a query's whole list, or one ISLAND of it (below); pdelta = where the run starts inside its
// query's list (predecessor indices are stored relative to the query)
template <int R, bool SKIP>
__device__ __forceinline__ void d_chain_run(const uint64_t *__restrict__ keys, const int64_t base, const int n, const int pdelta,
                                            const ChainOpt &o, int32_t *__restrict__ f, int32_t *__restrict__ p)
{
    const int lane = threadIdx.x & 63;
    const uint64_t *a = keys + base;
    const uint32_t max_gap = (uint32_t)o.max_gap, ddc = o.bw < o.max_gap ? (uint32_t)(o.max_gap - 1 - o.bw) : 0u;
    const uint32_t gap_q8 = (uint32_t)o.chain_gap_q8, skip_q8 = (uint32_t)o.chain_skip_q8;
    // fields of the anchors this lane owns: reference word (strand in bit 31), query position, span - 1, and the DP state.
    // An empty slot (key 0) has qi = 0, which fails the range test of dq against every j.
    uint32_t gi[R], qi[R], sp1[R]; int32_t B[R], bp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = r * 64 + lane;
        const uint64_t k = i < n ? a[i] : 0;
        gi[r] = (uint32_t)(k >> 32); qi[r] = (uint32_t)A_Q(k); sp1[r] = (uint32_t)A_SPAN(k) - 1u; B[r] = 2 * A_SPAN(k) + 1; bp[r] = -1;
    }
    uint32_t gcur = (uint32_t)__builtin_amdgcn_readlane((int)gi[0], 0);      // reference word of the anchor whose turn it is
    for (int jb = 0; jb < n; jb += 64 * R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j0 = jb + r * 64;
            if (j0 >= n) break;
            // the anchor that replaces this slot once its own is final
            const int inext = j0 + 64 * R + lane;
            const uint64_t knext = inext < n ? a[inext] : 0;
            const uint32_t ng = (uint32_t)(knext >> 32), nqp = (uint32_t)A_Q(knext), nsp1 = (uint32_t)A_SPAN(knext) - 1u;
            const int32_t nB = 2 * A_SPAN(knext) + 1;
            int32_t myB = 0, myp = -1;
            const int jn = n - j0 < 64 ? n - j0 : 64;
            for (int jj = 0; jj < jn; ++jj) {
                const int j = j0 + jj;
                const uint32_t gj = gcur, gj1 = gj + 1u;            // broadcast as the previous trip's "next anchor"
                // reference word of anchor j + 1 (still in its slot: a slot is refilled only when its own anchor is done)
                const uint32_t gnext = jj < 63 ? (uint32_t)__builtin_amdgcn_readlane((int)gi[r], jj + 1) : (uint32_t)__builtin_amdgcn_readlane((int)gi[(r + 1) % R], 0);
                // the owner of j keeps its final state for the store and takes its next anchor (selects, not a branch;
                // bp of a slot is only meaningful once B is even, so it is not reset)
                gcur = gnext;
                const bool me = lane == jj;
                myB = me ? B[r] : myB; myp = me ? bp[r] : myp;
                gi[r] = me ? ng : gi[r]; sp1[r] = me ? nsp1 : sp1[r];
                // Anchors are sorted by (strand, reference position): when even the NEXT anchor lies more than max_gap beyond j
                // (or on the other strand: bit 31), no later anchor can take j as a predecessor -- a scalar test that skips the
                // whole push for the isolated hits that repeats and random k-mer matches scatter over the genome (their query
                // position and score are not even broadcast).
                if (gnext - gj > max_gap) { qi[r] = me ? nqp : qi[r]; B[r] = me ? nB : B[r]; continue; }
                const uint32_t qj1 = (uint32_t)__builtin_amdgcn_readlane((int)qi[r], jj) + 1u;
                const int32_t fj2p2 = (__builtin_amdgcn_readlane(B[r], jj) & ~1) + 2;
                qi[r] = me ? nqp : qi[r]; B[r] = me ? nB : B[r];
#pragma unroll
                for (int s = 0; s < R; ++s)     // anchors currently owned: index > j and <= j + 64R by construction
                    d_chain_push<SKIP>(gi[s], qi[s], sp1[s], gj1, qj1, fj2p2, j, max_gap, ddc, gap_q8, skip_q8, B[s], bp[s]);
            }
            if (lane < jn) { f[base + j0 + lane] = myB >> 1; p[base + j0 + lane] = (myB & 1) ? -1 : myp + pdelta; }
        }
    }
}

// LAZY FAR LOOK-BACK (round 5; look-backs of 128 / 256 anchors).  The push loop above spends 19 instructions on every one of the H
// links of an anchor, and on a true chain -- anchors every few bases, f growing with every one of them -- the far ones never matter: the
// best predecessor is among the nearest.  Here an anchor takes pushes from its 64 NEAREST predecessors only (one slot, the R = 1 loop),
// and the H - 64 far ones are looked at only when they could still win:  a link scores at most span_i + f_j, so with
//   M_i = max f over the far predecessors of i     (lane l of chunk c: lanes >= l of chunk c-R, all of the chunks between, lanes < l of
//                                                   chunk c-1: one suffix maximum, the chunks' maxima, one prefix maximum -- two scans
//                                                   per finished chunk, kept in registers)
// the far ones are out as soon as span_i + M_i <= best so far (ties go to the LARGER index, i.e. to the near ones).  The order of the
// look-back does not matter for the result: f(i) = the maximum, p(i) = the largest index attaining it provided it beats span_i -- what
// the ascending loop with ">=" yields (d_chain_push).  When the bound does not decide (anchors without a near predecessor: chain starts,
// repeats) a scalar test on the nearest far predecessor's position (index i - 65: everything further back lies even further away) sorts
// out the isolated ones, and what is left evaluates its far links on the spot, one per lane and chunk, from the last R chunks'
// (reference word, query position, f) kept in registers, with a wave reduction (maximum, then the largest index among its holders).
// Same f, same p as d_chain_run (tests/test_gpu_parity.py compares both with the oracle; TELR_AB=chain_push keeps the full push loop).
template <int R, bool SKIP>
__device__ __forceinline__ void d_chain_run_lazy(const uint64_t *__restrict__ keys, const int64_t base, const int n, const int pdelta,
                                                 const ChainOpt &o, int32_t *__restrict__ f, int32_t *__restrict__ p)
{
    static_assert(R >= 2, "a look-back of 64 has no far part");
    constexpr int32_t NEG = -(1 << 29);
    const int lane = threadIdx.x & 63;
    const uint64_t *a = keys + base;
    const uint32_t max_gap = (uint32_t)o.max_gap, ddc = o.bw < o.max_gap ? (uint32_t)(o.max_gap - 1 - o.bw) : 0u;
    const uint32_t gap_q8 = (uint32_t)o.chain_gap_q8, skip_q8 = (uint32_t)o.chain_skip_q8;
    uint32_t gi, qi, sp1; int32_t B, bp;
    { const uint64_t k = lane < n ? a[lane] : 0; gi = (uint32_t)(k >> 32); qi = (uint32_t)A_Q(k); sp1 = (uint32_t)A_SPAN(k) - 1u; B = 2 * A_SPAN(k) + 1; bp = -1; }
    // the last R finished chunks, newest first: reference word, query position (0x7fffffff: no such anchor -- fails every range test), 2 f + 2;
    // their suffix maxima of f by lane, the newest one's prefix maximum, every chunk's maximum
    uint32_t hg[R], hq[R]; int32_t hf[R], hsuf[R], hfull[R], pre0 = NEG;
#pragma unroll
    for (int h = 0; h < R; ++h) { hg[h] = 0; hq[h] = 0x7fffffffu; hf[h] = 0; hsuf[h] = NEG; hfull[h] = NEG; }
    uint32_t gcur = (uint32_t)__builtin_amdgcn_readlane((int)gi, 0);
    for (int j0 = 0; j0 < n; j0 += 64) {
        // the far bound of this chunk's anchors (every lane holds its anchor of THIS chunk now)
        int32_t Ms = hsuf[R - 1] > pre0 ? hsuf[R - 1] : pre0;
#pragma unroll
        for (int h = 1; h < R - 1; ++h) Ms = hfull[h] > Ms ? hfull[h] : Ms;
        Ms += (int32_t)sp1 + 1;
        const int inext = j0 + 64 + lane;
        const uint64_t knext = inext < n ? a[inext] : 0;
        const uint32_t ng = (uint32_t)(knext >> 32), nqp = (uint32_t)A_Q(knext), nsp1 = (uint32_t)A_SPAN(knext) - 1u;
        const int32_t nB = 2 * A_SPAN(knext) + 1;
        int32_t myB = 0, myp = -1;
        const int jn = n - j0 < 64 ? n - j0 : 64;
        for (int jj = 0; jj < jn; ++jj) {
            const int j = j0 + jj;
            const uint32_t gj = gcur, gj1 = gj + 1u;
            const uint32_t gnext = jj < 63 ? (uint32_t)__builtin_amdgcn_readlane((int)gi, jj + 1) : (uint32_t)__builtin_amdgcn_readlane((int)ng, 0);
            gcur = gnext;
            int32_t sB = __builtin_amdgcn_readlane(B, jj);
            const int32_t sMs = __builtin_amdgcn_readlane(Ms, jj);
            const bool me = lane == jj;
            if (sMs > (sB >> 1)) {
                // the nearest far predecessor, index j - 65: lane jj - 1 of the newest chunk, or lane 63 of the one before
                const uint32_t gfar = jj ? (uint32_t)__builtin_amdgcn_readlane((int)hg[0], jj - 1) : (uint32_t)__builtin_amdgcn_readlane((int)hg[1], 63);
                if (gj - gfar <= max_gap) {
                    const uint32_t qS = (uint32_t)__builtin_amdgcn_readlane((int)qi, jj), spS = (uint32_t)__builtin_amdgcn_readlane((int)sp1, jj);
                    int32_t Bl = NEG * 2, bl = -1;
#pragma unroll
                    for (int h = R - 1; h >= 0; --h) {          // oldest chunk first: ascending predecessor index, ">=" keeps the largest
                        const bool in_set = h == 0 ? lane < jj : h == R - 1 ? lane >= jj : true;
                        int32_t Bt = Bl, bt = bl;
                        d_chain_push<SKIP>(gj, qS, spS, hg[h] + 1u, hq[h] + 1u, hf[h], j0 - 64 * (h + 1) + lane, max_gap, ddc, gap_q8, skip_q8, Bt, bt);
                        Bl = in_set ? Bt : Bl; bl = in_set ? bt : bl;
                    }
                    int32_t m = Bl;
#pragma unroll
                    for (int s = 32; s >= 1; s >>= 1) { const int32_t v = __shfl_xor(m, s); m = v > m ? v : m; }
                    if (m > sB) {           // strictly better than every near link (and than the span alone): the far link it is
                        int32_t w = Bl == m ? bl : -1;
#pragma unroll
                        for (int s = 32; s >= 1; s >>= 1) { const int32_t v = __shfl_xor(w, s); w = v > w ? v : w; }
                        B = me ? m : B; bp = me ? w : bp; sB = __builtin_amdgcn_readfirstlane(m);
                    }
                }
            }
            const uint32_t qj1 = (uint32_t)__builtin_amdgcn_readlane((int)qi, jj) + 1u;
            // lane jj hands over: its anchor's final state goes to (myB, myp), its slot takes the anchor 64 further on.  Six moves of ONE
            // lane under an execution mask of that lane (full-rate v_mov) instead of six v_cndmask_b32 with a 64-bit mask operand (VOP3,
            // half rate): the loop's bookkeeping, not its arithmetic, was a third of its issue cycles.  (The wave is whole here.)
            {
                const uint64_t one = 1ULL << jj; uint64_t sv;
                asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[m]\n\t"
                             "v_mov_b32 %[myB], %[B]\n\tv_mov_b32 %[myp], %[bp]\n\tv_mov_b32 %[gi], %[ng]\n\tv_mov_b32 %[sp1], %[nsp1]\n\t"
                             "v_mov_b32 %[qi], %[nqp]\n\tv_mov_b32 %[B], %[nB]\n\ts_mov_b64 exec, %[sv]"
                             : [myB] "+v"(myB), [myp] "+v"(myp), [gi] "+v"(gi), [sp1] "+v"(sp1), [qi] "+v"(qi), [B] "+v"(B), [sv] "=&s"(sv)
                             : [bp] "v"(bp), [ng] "v"(ng), [nsp1] "v"(nsp1), [nqp] "v"(nqp), [nB] "v"(nB), [m] "s"(one));
            }
            if (gnext - gj <= max_gap) {
                const int32_t fj2p2 = (sB & ~1) + 2;
                d_chain_push<SKIP>(gi, qi, sp1, gj1, qj1, fj2p2, j, max_gap, ddc, gap_q8, skip_q8, B, bp);
            }
        }
        if (lane < jn) { f[base + j0 + lane] = myB >> 1; p[base + j0 + lane] = (myB & 1) ? -1 : myp + pdelta; }
        if (j0 + 64 < n) {
            // this chunk becomes history (its anchors' words come back from memory: the slot holds the next chunk's by now)
#pragma unroll
            for (int h = R - 1; h >= 1; --h) { hg[h] = hg[h - 1]; hq[h] = hq[h - 1]; hf[h] = hf[h - 1]; hsuf[h] = hsuf[h - 1]; hfull[h] = hfull[h - 1]; }
            const uint64_t k = a[j0 + lane];
            hg[0] = (uint32_t)(k >> 32); hq[0] = (uint32_t)A_Q(k); hf[0] = (myB & ~1) + 2;
            const int32_t fv = myB >> 1;
            int32_t inc = fv, suf = fv;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const int32_t u = __shfl_up(inc, s), d = __shfl_down(suf, s);
                if (lane >= s) inc = u > inc ? u : inc;
                if (lane + s < 64) suf = d > suf ? d : suf;
            }
            const int32_t ex = __shfl_up(inc, 1);
            pre0 = lane ? ex : NEG; hsuf[0] = suf; hfull[0] = __builtin_amdgcn_readlane(suf, 0);
        }
    }
}

// WHICH LOOP (round 6).  The lazy look-back wins on true chains (the far links are never scored) and loses badly inside satellites and
// tandem arrays: there an anchor's collinear predecessor lies (copies in the read) x (a few bases) anchors back, past the 64 nearest,
// the bound never decides and EVERY anchor pays the far evaluation -- R gathered links plus two wave reductions, ~3 times the price
// of the plain push loop's R links -- and such a query is one wave of 10^5-10^6 anchors that the whole range waits for.  A run of
// at least `dense_n` anchors is sampled at 64 places: where 64 consecutive anchors sit within `dense_span` reference bases (several
// anchors per base: a lattice of repeat copies, not a chain) in a third of the samples, the run takes the push loop.  Same f, same p
// either way; TELR_AB=chain_lazy / chain_push pin one loop.
__device__ __forceinline__ bool d_chain_dense(const uint64_t *__restrict__ a, const int n, const ChainOpt &o)
{
    if (o.dense_n <= 0 || n < o.dense_n || n < 128) return false;
    const int i = (int)(((int64_t)(threadIdx.x & 63) * (n - 65)) / 63);
    const uint32_t g0 = (uint32_t)(a[i] >> 32), g1 = (uint32_t)(a[i + 64] >> 32);      // (another strand: bit 31 makes the difference huge)
    return __popcll(__ballot(g1 - g0 <= (uint32_t)o.dense_span)) >= 21;
}
// R WAVES ON ONE RUN (round 6).  The push loop's R links per anchor are independent: wave w of a workgroup of R owns the chunks
// c = w (mod R) of 64 anchors, one anchor per lane.  An anchor's look-back of 64 R predecessors is then
//   * lanes >= l of chunk c - R: the wave's OWN previous chunk -- a lane takes its next anchor the moment its own is final, as in the
//     push loop, and keeps receiving the rest of that chunk's pushes;
//   * chunks c - R + 1 .. c - 1: the other waves' -- read from a ring of the last 64 R final anchors in LDS (reference word + 1,
//     query position + 1, 2 f + 2), 64 entries at a time into registers, pushed with v_readlane, as far as the counter `done` says
//     they are published (a wave polls it with s_sleep: all R waves of a workgroup are resident, the wave that finalises the lowest
//     unfinished chunk never waits);
//   * lanes < l of chunk c: the wave's own 64-step loop, publishing eight anchors at a time.
// Pushes reach an anchor in ascending order of j, so ">=" keeps the largest index among ties: same f, same p as the other loops.
// An entry of the ring is overwritten by the anchor 64 R further on, which is published only after every chunk that reads the old
// one has finished absorbing.  The critical path per anchor is one link (~45 instructions) instead of R links.  Only for runs that a
// whole call would wait for (o.mw_n anchors: a window read across a satellite against its contig brings 10^6): thousands of dense
// runs keep the device busy by themselves, and workgroups of R waves for EVERY query cost the easy genome 5 % of its step (measured).
template <int R, bool SKIP>
__device__ __forceinline__ void d_chain_run_mw(const uint64_t *__restrict__ keys, const int64_t base, const int n, const int pdelta,
                                               const ChainOpt &o, int32_t *__restrict__ f, int32_t *__restrict__ p, uint4 *ring, int32_t *done_s)
{
    static_assert(R >= 2, "one wave is the push loop");
    constexpr int RING = 64 * R;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (the wave's number as a SCALAR: chunk bounds, loop control and ring offsets stay on the scalar side)
    const uint64_t *a = keys + base;
    const uint32_t max_gap = (uint32_t)o.max_gap, ddc = o.bw < o.max_gap ? (uint32_t)(o.max_gap - 1 - o.bw) : 0u;
    const uint32_t gap_q8 = (uint32_t)o.chain_gap_q8, skip_q8 = (uint32_t)o.chain_skip_q8;
    if (threadIdx.x == 0) *done_s = 0;
    __syncthreads();
    const int nchunk = (n + 63) >> 6;
    uint32_t gi, qi, sp1; int32_t B, bp = -1;
    { const int i = wv * 64 + lane; const uint64_t k = i < n ? a[i] : 0; gi = (uint32_t)(k >> 32); qi = (uint32_t)A_Q(k); sp1 = (uint32_t)A_SPAN(k) - 1u; B = 2 * A_SPAN(k) + 1; }
    for (int c = wv; c < nchunk; c += R) {
        const int j0 = c * 64;
        // the anchor that takes this lane's place in chunk c + R
        const int inext = j0 + RING + lane;
        const uint64_t knext = inext < n ? a[inext] : 0;
        const uint32_t ng = (uint32_t)(knext >> 32), nqp = (uint32_t)A_Q(knext), nsp1 = (uint32_t)A_SPAN(knext) - 1u;
        const int32_t nB = 2 * A_SPAN(knext) + 1;
        // ---- the other waves' chunks
        int j = j0 - 64 * (R - 1); if (j < 0) j = 0;
        int polls = 0;
        while (j < j0) {
            const int avail = __builtin_amdgcn_readfirstlane(__hip_atomic_load(done_s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (avail <= j) { if (++polls > (1 << 24)) break; __builtin_amdgcn_s_sleep(2); continue; }       // (the bound only keeps a broken build from hanging the device: ~1 s)
            int lim = avail < j0 ? avail : j0; if (lim > j + 64) lim = j + 64;
            const int cnt = lim - j;
            uint4 e = make_uint4(0u, 0u, 0u, 0u);
            if (lane < cnt) e = ring[(j + lane) & (RING - 1)];
            for (int t = 0; t < cnt; ++t) {
                const uint32_t gj1 = (uint32_t)__builtin_amdgcn_readlane((int)e.x, t), qj1 = (uint32_t)__builtin_amdgcn_readlane((int)e.y, t);
                const int32_t fj2p2 = __builtin_amdgcn_readlane((int)e.z, t);
                d_chain_push<SKIP>(gi, qi, sp1, gj1, qj1, fj2p2, j + t, max_gap, ddc, gap_q8, skip_q8, B, bp);
            }
            j = lim;
        }
        // ---- the wave's own chunk
        const int jn = n - j0 < 64 ? n - j0 : 64;
        int32_t myB = 0, myp = -1; uint32_t myg = 0, myq = 0;
        for (int jj = 0; jj < jn; ++jj) {
            const uint32_t gj1 = (uint32_t)__builtin_amdgcn_readlane((int)gi, jj) + 1u, qj1 = (uint32_t)__builtin_amdgcn_readlane((int)qi, jj) + 1u;
            const int32_t fj2p2 = (__builtin_amdgcn_readlane(B, jj) & ~1) + 2;
            // lane jj hands over (eight moves of ONE lane under an execution mask of that lane, as in the lazy loop; the wave is whole here)
            {
                const uint64_t one = 1ULL << jj; uint64_t sv;
                asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[m]\n\t"
                             "v_mov_b32 %[myB], %[B]\n\tv_mov_b32 %[myp], %[bp]\n\tv_mov_b32 %[myg], %[gi]\n\tv_mov_b32 %[myq], %[qi]\n\t"
                             "v_mov_b32 %[gi], %[ng]\n\tv_mov_b32 %[qi], %[nqp]\n\tv_mov_b32 %[sp1], %[nsp1]\n\tv_mov_b32 %[B], %[nB]\n\ts_mov_b64 exec, %[sv]"
                             : [myB] "+v"(myB), [myp] "+v"(myp), [myg] "+v"(myg), [myq] "+v"(myq), [gi] "+v"(gi), [qi] "+v"(qi), [sp1] "+v"(sp1), [B] "+v"(B), [sv] "=&s"(sv)
                             : [bp] "v"(bp), [ng] "v"(ng), [nqp] "v"(nqp), [nsp1] "v"(nsp1), [nB] "v"(nB), [m] "s"(one));
            }
            d_chain_push<SKIP>(gi, qi, sp1, gj1, qj1, fj2p2, j0 + jj, max_gap, ddc, gap_q8, skip_q8, B, bp);
            if ((jj & 7) == 7 || jj == jn - 1) {
                if (lane >= (jj & ~7) && lane <= jj) ring[(j0 + lane) & (RING - 1)] = make_uint4(myg + 1u, myq + 1u, (uint32_t)((myB & ~1) + 2), 0u);
                __hip_atomic_store(done_s, j0 + jj + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (lane < jn) { f[base + j0 + lane] = myB >> 1; p[base + j0 + lane] = (myB & 1) ? -1 : myp + pdelta; }
    }
}
// MODE 0: the push loop, 1: the lazy far look-back, 2: chosen per run.  `skip_mw`: runs that k_chain_mw takes (dense and at least
// o.mw_n anchors) are left to it.
__device__ __forceinline__ bool d_chain_is_mw(const bool dense, const int n, const ChainOpt &o) { return dense && o.mw_n > 0 && n >= o.mw_n; }
template <int R, bool SKIP, int MODE>
__device__ __forceinline__ void d_chain_any(const uint64_t *__restrict__ keys, const int64_t base, const int n, const int pdelta,
                                            const ChainOpt &o, int32_t *__restrict__ f, int32_t *__restrict__ p, const bool skip_mw)
{
    if constexpr (MODE == 0 || R < 2) d_chain_run<R, SKIP>(keys, base, n, pdelta, o, f, p);
    else if constexpr (MODE == 1) d_chain_run_lazy<R, SKIP>(keys, base, n, pdelta, o, f, p);
    else {
        const bool dense = d_chain_dense(keys + base, n, o);
        if (skip_mw && d_chain_is_mw(dense, n, o)) return;
        if (dense) d_chain_run<R, SKIP>(keys, base, n, pdelta, o, f, p);
        else d_chain_run_lazy<R, SKIP>(keys, base, n, pdelta, o, f, p);
    }
}
template <int R, bool SKIP, int MODE = 2>
__global__ void __launch_bounds__(64) k_chain(const uint64_t *__restrict__ keys, const int32_t *__restrict__ q_aoff, int32_t nq,
                                              ChainOpt o, int32_t *__restrict__ f, int32_t *__restrict__ p, const int32_t *__restrict__ q_order, int32_t skip_mw)
{
    if ((int)blockIdx.x >= nq) return;
    const int q = q_order ? q_order[blockIdx.x] : blockIdx.x;    // longest reads first: the kernel ends with the short ones
    d_chain_any<R, SKIP, MODE>(keys, q_aoff[q], q_aoff[q + 1] - q_aoff[q], 0, o, f, p, skip_mw != 0);
}
// the long dense runs of a call, R waves each (beside k_chain, on a stream of its own): `list` = the range's over-size queries
// (every run of o.mw_n > SEGSORT_CAP anchors is one of them), or every query when list == nullptr (tests with small thresholds)
template <int R, bool SKIP>
__global__ void __launch_bounds__(64 * R) k_chain_mw(const uint64_t *__restrict__ keys, const int32_t *__restrict__ q_aoff, int32_t nq, const int32_t *__restrict__ list,
                                                     ChainOpt o, int32_t *__restrict__ f, int32_t *__restrict__ p)
{
    __shared__ uint4 ring[64 * R];
    __shared__ int32_t done_s;
    if ((int)blockIdx.x >= nq) return;
    const int q = list ? list[blockIdx.x] : (int)blockIdx.x;
    const int64_t base = q_aoff[q]; const int n = q_aoff[q + 1] - q_aoff[q];
    if (!d_chain_is_mw(d_chain_dense(keys + base, n, o), n, o)) return;      // (the same test in every wave of the workgroup)
    d_chain_run_mw<R, SKIP>(keys, base, n, 0, o, f, p, ring, &done_s);
}
// ISLANDS.  Anchors are sorted by (strand, reference position) and a link needs 0 < dr <= max_gap, so wherever two consecutive
// anchors of a query lie more than max_gap apart (or on different strands: bit 31) no link crosses: the list falls into islands
// that chain independently -- same f, same p, the look-back window being a window of indices inside the island anyway.  One wave
// per query is the right grain for stage 1 (458,550 reads); a call with FEW queries and long lists -- the TE library against a
// thousand contigs (S5: 127 queries, ~100 k anchors each), the ALT sequences, the flanks -- left the device to 127 waves, each
// walking its list serially.  Such calls (nq <= CHAIN_ISL_NQ) chain island by island: k_isl_heads marks the island heads,
// a scan numbers them, k_isl_fill writes their offsets, and k_chain_isl is a grid-stride loop over the count the device holds.
#define CHAIN_ISL_NQ 4096
__global__ void __launch_bounds__(256) k_isl_heads(const uint64_t *__restrict__ keys, const int32_t *__restrict__ q_aoff, int32_t nq, uint32_t max_gap, int32_t *__restrict__ head)
{
    const int q = blockIdx.x;
    if (q >= nq) return;
    const int64_t base = q_aoff[q]; const int n = q_aoff[q + 1] - q_aoff[q];
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        head[base + i] = i == 0 || (uint32_t)(keys[base + i] >> 32) - (uint32_t)(keys[base + i - 1] >> 32) > max_gap;
}
__global__ void __launch_bounds__(256) k_isl_fill(const int32_t *__restrict__ q_aoff, int32_t nq, const int32_t *__restrict__ head, const int32_t *__restrict__ rank, int32_t na,
                                                  int32_t *__restrict__ isl_off, int32_t *__restrict__ isl_pd)
{
    const int q = blockIdx.x;
    if (q >= nq) return;
    const int64_t base = q_aoff[q]; const int n = q_aoff[q + 1] - q_aoff[q];
    for (int i = threadIdx.x; i < n; i += blockDim.x) if (head[base + i]) { const int id = rank[base + i]; isl_off[id] = (int32_t)base + i; isl_pd[id] = i; }
    if (q == 0 && threadIdx.x == 0) isl_off[rank[na]] = na;
}
template <int R, bool SKIP, int MODE = 2>
__global__ void __launch_bounds__(64) k_chain_isl(const uint64_t *__restrict__ keys, const int32_t *__restrict__ isl_off, const int32_t *__restrict__ isl_pd, const int32_t *__restrict__ nisl,
                                                  ChainOpt o, int32_t *__restrict__ f, int32_t *__restrict__ p)
{
    const int n_isl = *nisl;
    for (int s = blockIdx.x; s < n_isl; s += gridDim.x)
        d_chain_any<R, SKIP, MODE>(keys, isl_off[s], isl_off[s + 1] - isl_off[s], isl_pd[s], o, f, p, false);
}

// peaks: anchors with no successor of larger f
__global__ void k_nonpeak(const int32_t *__restrict__ q_aoff, const int32_t *__restrict__ f, const int32_t *__restrict__ p, uint8_t *__restrict__ nonpeak)
{
    const int q = blockIdx.x;
    const int64_t base = q_aoff[q]; const int n = q_aoff[q + 1] - q_aoff[q];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int pj = p[base + i];
        if (pj >= 0 && f[base + i] > f[base + pj]) nonpeak[base + pj] = 1;
    }
}
__global__ void k_peaks(const int32_t *__restrict__ q_aoff, const int32_t *__restrict__ f, const uint8_t *__restrict__ nonpeak,
                        int32_t min_sc, uint64_t *__restrict__ pk, int32_t *__restrict__ n_peaks, int32_t *__restrict__ pk_end)
{
    __shared__ int32_t cnt;
    const int q = blockIdx.x;
    const int64_t base = q_aoff[q]; const int n = q_aoff[q + 1] - q_aoff[q];
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (!nonpeak[base + i] && f[base + i] >= min_sc) {
            int s = atomicAdd(&cnt, 1);
            pk[base + s] = (uint64_t)(uint32_t)(0x7fffffff - f[base + i]) << 32 | (uint32_t)i;   // ascending == (f desc, i asc)
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { n_peaks[q] = cnt; pk_end[q] = (int32_t)base + cnt; }
}

struct ChainRec { int32_t score, cnt, a_off, pad; uint64_t a0, a1; };   // 32 B


// ---- back-tracking without a walker (round 2).  The predecessor links form a forest (p[i] < i, i - p[i] <= look-back).
// Visiting the peaks in rank order (f descending, index ascending) and walking up until a visited anchor is the same as:
//   owner(i) = the best-ranked peak in the subtree of i  (the first walker to reach i);
//   chain t  = { i : owner(i) = t }, a path from the peak of rank t up to the first anchor owned by a better peak
//              (it exists iff the peak owns itself); its anchors in index order are the chain in ascending order;
//   depth(i) = number of same-owner ancestors = position of i on its chain;  count = depth(peak) + 1;
//   score    = f(peak) - f(parent of the chain's top anchor)  (0 when the top is a root).
// owner is a reverse sweep (children push the minimum to parents), depth a forward sweep (list ranking); both run over
// blocks of 64 anchors with the link targets of the last `look-back` anchors in an LDS ring, and resolve the links that
// stay inside a block by six rounds of pointer jumping -- every lane works, where the walker kept 63 of 64 idle at LDS
// latency per anchor.  One wave per query whatever its size (no LDS tiers, no lists).
#define BT_INF 0xffffffffu
#define BT_RING 512                 // >= look-back (<= 256) + 2 x 64
__global__ void k_bt_rank(const int32_t *__restrict__ q_aoff, const uint64_t *__restrict__ pk, const int32_t *__restrict__ n_peaks, uint32_t *__restrict__ owner)
{
    const int q = blockIdx.x;
    const int64_t base = q_aoff[q]; const int np = n_peaks[q];
    for (int t = threadIdx.x; t < np; t += blockDim.x) owner[base + (uint32_t)(pk[base + t] & 0xffffffffu)] = (uint32_t)t;
}
__global__ void __launch_bounds__(64) k_bt_owner(const int32_t *__restrict__ q_aoff, int32_t nq, const int32_t *__restrict__ p, int32_t lookback,
                                                 uint32_t *__restrict__ owner, const int32_t *__restrict__ q_order)
{
    __shared__ uint32_t W[BT_RING];
    __shared__ int32_t UP[64];
    const int q = q_order ? q_order[blockIdx.x] : blockIdx.x, lane = threadIdx.x;
    const int64_t base = q_aoff[q]; const int n = q_aoff[q + 1] - q_aoff[q];
    if (n <= 0) return;
    const int nblk = (n + 63) >> 6;
    for (int blk = nblk - 1; blk >= 0; --blk) {
        const int b0 = blk << 6;
        // ring entries that enter the window [b0 - lookback, b0 + 64): all of it for the first block handled, the lowest 64 afterwards
        if (blk == nblk - 1) { for (int x = b0 - lookback + lane; x < b0 + 64; x += 64) if (x >= 0 && x < n) W[x & (BT_RING - 1)] = owner[base + x]; }
        else { const int x = b0 - lookback + lane; if (x >= 0) W[x & (BT_RING - 1)] = owner[base + x]; }
        const int i = b0 + lane; const bool valid = i < n;
        const int pi = valid ? p[base + i] : -1;
        int up = pi >= b0 ? pi - b0 : -1;
        UP[lane] = up;
        for (int k = 0; k < 6; ++k) {
            if (__ballot(up >= 0) == 0) break;
            if (up >= 0) atomicMin(&W[(b0 + up) & (BT_RING - 1)], W[i & (BT_RING - 1)]);
            const int nup = up >= 0 ? UP[up] : -1;
            UP[lane] = nup; up = nup;
        }
        if (valid) {
            const uint32_t v = W[i & (BT_RING - 1)];
            owner[base + i] = v;
            if (pi >= 0 && pi < b0) atomicMin(&W[pi & (BT_RING - 1)], v);
        }
    }
}
__global__ void __launch_bounds__(64) k_bt_depth(const int32_t *__restrict__ q_aoff, int32_t nq, const int32_t *__restrict__ p, const uint32_t *__restrict__ owner,
                                                 int32_t *__restrict__ depth, int32_t *__restrict__ ch_top, const int32_t *__restrict__ q_order)
{
    __shared__ uint32_t OWN[BT_RING];
    __shared__ int32_t DEP[BT_RING];
    __shared__ int32_t UP[64], VAL[64];
    const int q = q_order ? q_order[blockIdx.x] : blockIdx.x, lane = threadIdx.x;
    const int64_t base = q_aoff[q]; const int n = q_aoff[q + 1] - q_aoff[q];
    const int nblk = (n + 63) >> 6;
    for (int blk = 0; blk < nblk; ++blk) {
        const int b0 = blk << 6, i = b0 + lane; const bool valid = i < n;
        const uint32_t o = valid ? owner[base + i] : BT_INF;
        const int pi = valid ? p[base + i] : -1;
        OWN[i & (BT_RING - 1)] = o;
        const uint32_t po = pi >= 0 ? OWN[pi & (BT_RING - 1)] : BT_INF;
        const bool conn = o != BT_INF && pi >= 0 && po == o;           // the parent is on the same chain
        int up = -1, val = 0;
        if (conn) { if (pi >= b0) { up = pi - b0; val = 1; } else val = DEP[pi & (BT_RING - 1)] + 1; }
        UP[lane] = up; VAL[lane] = val;
        for (int k = 0; k < 6; ++k) {
            if (__ballot(up >= 0) == 0) break;
            const int add = up >= 0 ? VAL[up] : 0, nup = up >= 0 ? UP[up] : -1;
            val += add;
            VAL[lane] = val; UP[lane] = nup; up = nup;
        }
        DEP[i & (BT_RING - 1)] = val;
        if (valid) {
            depth[base + i] = val;
            if (o != BT_INF && !conn) ch_top[base + o] = i;            // exactly one top anchor per chain
        }
    }
}
// chains of a query in rank order: which exist, pass the filters, where their anchors go
__global__ void __launch_bounds__(64) k_bt_emit(const uint64_t *__restrict__ keys, const int32_t *__restrict__ q_aoff, int32_t nq, const int32_t *__restrict__ f,
                                                const int32_t *__restrict__ p, const uint64_t *__restrict__ pk, const int32_t *__restrict__ n_peaks,
                                                const int32_t *__restrict__ ch_off, int32_t min_sc, int32_t min_cnt, const uint32_t *__restrict__ owner,
                                                const int32_t *__restrict__ depth, const int32_t *__restrict__ ch_top, int32_t *__restrict__ ch_aoff,
                                                ChainRec *__restrict__ rec, int32_t *__restrict__ n_chains, const int32_t *__restrict__ q_order)
{
    const int q = q_order ? q_order[blockIdx.x] : blockIdx.x, lane = threadIdx.x;
    const int64_t base = q_aoff[q]; const int np = n_peaks[q];
    ChainRec *out = rec + ch_off[q];
    int nch = 0, wr = 0;
    for (int t0 = 0; t0 < np; t0 += 64) {
        const int t = t0 + lane;
        bool keep = false; int cnt = 0, sc = 0, idx = 0, top = 0;
        if (t < np) {
            const uint64_t key = pk[base + t];
            idx = (int)(uint32_t)(key & 0xffffffffu);
            const int fi = 0x7fffffff - (int)(uint32_t)(key >> 32);
            if (owner[base + idx] == (uint32_t)t) {
                cnt = depth[base + idx] + 1;
                top = ch_top[base + t];
                const int pj = p[base + top];
                sc = fi - (pj >= 0 ? f[base + pj] : 0);
                keep = sc >= min_sc && cnt >= min_cnt;
            }
        }
        const uint64_t km = __ballot(keep);
        const int before = __popcll(km & ((1ULL << lane) - 1));
        int ps = keep ? cnt : 0;                       // inclusive scan of the kept counts
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(ps, d); if (lane >= d) ps += v; }
        if (t < np) ch_aoff[base + t] = keep ? wr + ps - cnt : -1;
        if (keep) {
            ChainRec r; r.score = sc; r.cnt = cnt; r.a_off = wr + ps - cnt; r.pad = 0;
            r.a0 = keys[base + top]; r.a1 = keys[base + idx];
            out[nch + before] = r;
        }
        nch += __popcll(km); wr += __shfl(ps, 63);
    }
    if (lane == 0) n_chains[q] = nch;
}
__global__ void k_bt_scatter(const uint64_t *__restrict__ keys, const int32_t *__restrict__ q_aoff, const uint32_t *__restrict__ owner, const int32_t *__restrict__ depth,
                             const int32_t *__restrict__ ch_aoff, uint64_t *__restrict__ canch, const int32_t *__restrict__ q_order)
{
    const int q = q_order ? q_order[blockIdx.x] : blockIdx.x;
    const int64_t base = q_aoff[q]; const int n = q_aoff[q + 1] - q_aoff[q];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t o = owner[base + i];
        if (o == BT_INF) continue;
        const int ao = ch_aoff[base + o];
        if (ao >= 0) canch[base + ao + depth[base + i]] = keys[base + i];
    }
}

// ---------------------------------------------------------------------------------------
// 5. DP problems
struct KeptChain {          // uploaded by the host after chain selection
    int32_t qid, tid, rev, cnt;
    int64_t a_glob;         // offset of the chain's anchors in canch
    int32_t rs, qs, re, qe; // chain box: target-local / strand-adjusted query
    int32_t qlen, tlen;
    int64_t qbase, tbase;   // packed base offsets of query / target
    uint32_t goff; int32_t pad;
};

// ---- chain selection, pass 1, on the device (round 2; the host did this between two synchronisations: 125 MB of chain
// records over PCIe and ~7 ms of host work per Gbp of reads).  Same outcome as select_chains() of the oracle:
// chains of a query in (score desc, discovery asc) order; a chain overlapping a better PRIMARY by more than
// mask_level x the shorter of the two query intervals is its secondary; secondaries survive with score >= pri_ratio x
// parent and while fewer than best_n are kept (per target with TELR_MF_PER_TARGET).  One wave per query walks the sorted
// chains; the lanes test one chain against 64 primaries at a time.  The float products / compares are the host's.
struct SelOpt { float mask_level, pri_ratio; int32_t best_n, secondary, per_target; };
struct KeptLite { int32_t qid, score, cnt, rev, tid, rs, re, qs, qe, pad; };      // what the host still needs of a kept chain (40 B)
__global__ void k_sel_keys(const int32_t *__restrict__ ch_off, const int32_t *__restrict__ n_chains, const ChainRec *__restrict__ rec,
                           uint64_t *__restrict__ key, int32_t *__restrict__ seg_end)
{
    const int q = blockIdx.x;
    const int base = ch_off[q], n = n_chains[q];
    for (int c = threadIdx.x; c < n; c += blockDim.x) key[base + c] = (uint64_t)(uint32_t)(0x7fffffff - rec[base + c].score) << 32 | (uint32_t)c;
    if (threadIdx.x == 0) seg_end[q] = base + n;
}
// box of a chain from its first / last anchor (the host's HostChain)
__device__ __forceinline__ void d_chain_box(const ChainRec &r, const uint32_t *__restrict__ goff, int n_targets, int &rev, int &tid, int &rs, int &re, int &qs, int &qe)
{
    rev = (int)(r.a0 >> 63);
    tid = d_tid_of(goff, n_targets, (uint32_t)A_G(r.a0));
    const int go = (int)goff[tid];
    rs = A_G(r.a0) - go - A_SPAN(r.a0) + 1; re = A_G(r.a1) - go + 1;
    if (rs < 0) rs = 0;          // the span is the query minimizer's: with HPC the target's copy may be shorter (minimap2 clamps the same way)
    qs = A_Q(r.a0) - A_SPAN(r.a0) + 1;      qe = A_Q(r.a1) + 1;
}
// The primaries found so far (query interval, target, score) live in LDS (SEL_PCAP of them; beyond that in the global
// scratch regions [ch_off[q], ...) pfs/pfe/ptid/pkey), the kept-secondary tallies per target of the per-target mode in
// tcnt_t/tcnt_n (global, same regions).  64 chains are fetched and boxed at a time (one per lane), then walked in order
// with lane broadcasts, so the sequential loop touches no global memory in the common case.  keep = flag per SORTED position.
#define SEL_PCAP 1024
__global__ void __launch_bounds__(64) k_select1(const int32_t *__restrict__ ch_off, const int32_t *__restrict__ n_chains, const ChainRec *__restrict__ rec,
                                                const uint64_t *__restrict__ skey, const int32_t *__restrict__ qlen_of, const uint32_t *__restrict__ goff, int32_t n_targets,
                                                SelOpt o, int32_t *__restrict__ pfs, int32_t *__restrict__ pfe, int32_t *__restrict__ ptid, int32_t *__restrict__ pkey,
                                                int32_t *__restrict__ tcnt_t, int32_t *__restrict__ tcnt_n, uint8_t *__restrict__ keep, int32_t *__restrict__ n_kept,
                                                const int32_t *__restrict__ q_order)
{
    __shared__ int32_t Lfs[SEL_PCAP], Lfe[SEL_PCAP], Ltid[SEL_PCAP], Lkey[SEL_PCAP];
    const int q = q_order ? q_order[blockIdx.x] : blockIdx.x, lane = threadIdx.x;
    const int base = ch_off[q], n = n_chains[q], qlen = qlen_of[q];
    int n_prim = 0, n2_all = 0, n_t = 0, nk = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int cnt = n - i0 < 64 ? n - i0 : 64;
        int my_fs = 0, my_fe = 0, my_tid = 0, my_sc = 0;
        if (lane < cnt) {
            const ChainRec r = rec[base + (int)(uint32_t)(skey[base + i0 + lane] & 0xffffffffu)];
            int rev, rs, re, qs, qe;
            d_chain_box(r, goff, n_targets, rev, my_tid, rs, re, qs, qe);
            my_fs = rev ? qlen - qe : qs; my_fe = rev ? qlen - qs : qe; my_sc = r.score;
        }
        bool my_keep = false;
        for (int u = 0; u < cnt; ++u) {
            const int fs = __builtin_amdgcn_readlane(my_fs, u), fe = __builtin_amdgcn_readlane(my_fe, u), tid = __builtin_amdgcn_readlane(my_tid, u),
                      sc = __builtin_amdgcn_readlane(my_sc, u);
            int parent = -1;
            for (int c0 = 0; c0 < n_prim && parent < 0; c0 += 64) {
                const int j = c0 + lane;
                bool hit = false;
                if (j < n_prim) {
                    const int jt = j < SEL_PCAP ? Ltid[j] : ptid[base + j];
                    if (!o.per_target || jt == tid) {
                        const int js = j < SEL_PCAP ? Lfs[j] : pfs[base + j], je = j < SEL_PCAP ? Lfe[j] : pfe[base + j];
                        const int lo = fs > js ? fs : js, hi = fe < je ? fe : je;
                        const int ol = hi > lo ? hi - lo : 0;
                        const int mn = (fe - fs) < (je - js) ? (fe - fs) : (je - js);
                        hit = (float)ol > o.mask_level * (float)mn;
                    }
                }
                const uint64_t m = __ballot(hit);
                if (m) parent = c0 + __ffsll((unsigned long long)m) - 1;
            }
            bool kp;
            if (parent < 0) {
                if (lane == 0) {
                    if (n_prim < SEL_PCAP) { Lfs[n_prim] = fs; Lfe[n_prim] = fe; Ltid[n_prim] = tid; Lkey[n_prim] = sc; }
                    else { pfs[base + n_prim] = fs; pfe[base + n_prim] = fe; ptid[base + n_prim] = tid; pkey[base + n_prim] = sc; }
                }
                if (n_prim >= SEL_PCAP) __threadfence_block();
                ++n_prim; kp = true;
            } else {
                kp = false;
                const int pk_ = parent < SEL_PCAP ? Lkey[parent] : pkey[base + parent];
                if (o.secondary && !((float)sc < (float)pk_ * o.pri_ratio)) {
                    if (!o.per_target) { if (n2_all < o.best_n) { kp = true; ++n2_all; } }
                    else {
                        int z = -1;
                        for (int c0 = 0; c0 < n_t && z < 0; c0 += 64) {
                            const int j = c0 + lane;
                            const uint64_t m = __ballot(j < n_t && tcnt_t[base + j] == tid);
                            if (m) z = c0 + __ffsll((unsigned long long)m) - 1;
                        }
                        if (z < 0) { z = n_t++; if (lane == 0) { tcnt_t[base + z] = tid; tcnt_n[base + z] = 0; } __threadfence_block(); }
                        const int cur = tcnt_n[base + z];
                        if (cur < o.best_n) { kp = true; if (lane == 0) tcnt_n[base + z] = cur + 1; __threadfence_block(); }
                    }
                }
            }
            if (lane == u) my_keep = kp;
            nk += kp ? 1 : 0;
        }
        if (lane < cnt) keep[base + i0 + lane] = my_keep ? 1 : 0;
    }
    if (lane == 0) n_kept[q] = nk;
}
__global__ void __launch_bounds__(64) k_select1_write(const int32_t *__restrict__ ch_off, const int32_t *__restrict__ n_chains, const ChainRec *__restrict__ rec,
                                                      const uint64_t *__restrict__ skey, const uint8_t *__restrict__ keep, const int32_t *__restrict__ k_off,
                                                      const int32_t *__restrict__ q_aoff, int32_t q0, const int32_t *__restrict__ qlen_of, const int64_t *__restrict__ qboff,
                                                      const uint32_t *__restrict__ goff, const int32_t *__restrict__ tlen_of, const int64_t *__restrict__ tboff, int32_t n_targets,
                                                      KeptChain *__restrict__ kc, KeptLite *__restrict__ kl)
{
    const int q = blockIdx.x, lane = threadIdx.x;
    const int base = ch_off[q], n = n_chains[q];
    int w = k_off[q];
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool kp = i < n && keep[base + i];
        const uint64_t m = __ballot(kp);
        if (kp) {
            const int x = w + __popcll(m & ((1ULL << lane) - 1));
            const ChainRec r = rec[base + (int)(uint32_t)(skey[base + i] & 0xffffffffu)];
            int rev, tid, rs, re, qs, qe;
            d_chain_box(r, goff, n_targets, rev, tid, rs, re, qs, qe);
            KeptChain K; K.qid = q0 + q; K.tid = tid; K.rev = rev; K.cnt = r.cnt; K.a_glob = (int64_t)q_aoff[q] + r.a_off; K.rs = rs; K.qs = qs; K.re = re; K.qe = qe;
            K.qlen = qlen_of[q]; K.tlen = tlen_of[tid]; K.qbase = qboff[q]; K.tbase = tboff[tid]; K.goff = goff[tid]; K.pad = 0;
            kc[x] = K;
            KeptLite L; L.qid = q0 + q; L.score = r.score; L.cnt = r.cnt; L.rev = rev; L.tid = tid; L.rs = rs; L.re = re; L.qs = qs; L.qe = qe; L.pad = 0;
            kl[x] = L;
        }
        w += __popcll(m);
    }
}
struct DpProb {             // 64 B
    int64_t qi0, ti0;       // absolute packed base index of DP base 0
    int64_t tb_off;         // byte offset into the trace-back scratch
    int64_t cig_off;        // op offset into the raw cigar scratch
    int32_t m, n, dlo, dhi;
    int8_t qstep, tstep, qcomp, kind;   // kind 0 fill, 1 left ext, 2 right ext, 3 diagonal fallback
    int32_t chain;          // kept-chain index
    int32_t pad[2];
};
struct DpRes { int32_t score, bi, bj, nops, mlen, cells, tbases, mcols; };   // 32 B; mcols = M columns of the path

// first pass: 2 + q4*floor(sqrt(min(m,n)))/16 diagonals of slack; retry (path touched a band edge): the wide band,
// kept within 1024 diagonals where the first-pass width allows it (oracle: fill_band / fill_band_wide)
#define ADAPT_MAX_STEPS 1000       // m+n above which a segment skips the narrow pass
__host__ __device__ __forceinline__ int d_isqrt32(int v)      // exact floor(sqrt(v)), v >= 0
{
    int r = (int)sqrtf((float)v);
    while ((int64_t)r * r > v) --r;
    while ((int64_t)(r + 1) * (r + 1) <= v) ++r;
    return r;
}
__host__ __device__ __forceinline__ int d_fill_band(int m, int n, int bw, int q4)
{
    const int mn = m < n ? m : n, W = 2 + (((q4 > 0 ? q4 : 8) * d_isqrt32(mn)) >> 4);
    return W < bw ? W : bw;
}
__host__ __device__ __forceinline__ int d_fill_band_wide(int m, int n, int bw, int q4)
{
    const int mn = m < n ? m : n, dl = n - m, adl = dl < 0 ? -dl : dl;
    int W = mn <= 512 ? 24 + (mn >> 3) : 88 + ((mn - 512) >> 4);
    if (W > bw) W = bw;
    int cap = (1022 - adl) / 2; const int Wn = d_fill_band(m, n, bw, q4);
    if (cap < Wn) cap = Wn;
    return W < cap ? W : cap;
}
__device__ __forceinline__ int d_even_lo(int lo) { return lo - (lo & 1); }

// DP problems of a kept chain (oracle align_chain): PASS 0 counts them, PASS 1 writes the descriptors.  One wave per
// chain: a window of 64 consecutive anchors per iteration, the greedy cuts of the window found with ballots (the cut condition is monotone along a chain), and
// the descriptors of the window's cuts built and written by as many lanes in parallel.
template <int PASS>
__global__ void __launch_bounds__(64) k_segments_w(const KeptChain *__restrict__ kc, int32_t nk, const uint64_t *__restrict__ canch,
                                                   int32_t min_ksw_len, int32_t bw, int32_t band_q4, int32_t ext_max, int32_t ext_band, int32_t bw_long,
                                                   int32_t *__restrict__ nprob, const int32_t *__restrict__ prob_off, DpProb *__restrict__ probs)
{
    __shared__ int32_t cut_r[64], cut_q[64], prev_r[64], prev_q[64];
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c >= nk) return;
    const KeptChain K = kc[c];
    const uint64_t *ca = canch + K.a_glob;
    int np = 0;
    DpProb *out = PASS ? probs + prob_off[c] : nullptr;
    const int go = (int)K.goff;
    if (K.qs > 0 && K.rs > 0) {          // left extension
        if (PASS && lane == 0) {
            int mq = K.qs < ext_max ? K.qs : ext_max, mt = K.rs < mq + ext_band ? K.rs : mq + ext_band;
            DpProb P; P.m = mq; P.n = mt; P.dlo = d_even_lo(-ext_band); P.dhi = ext_band; P.kind = 1; P.chain = c;
            P.tstep = -1; P.ti0 = K.tbase + K.rs - 1; P.qcomp = (int8_t)K.rev;
            if (K.rev) { P.qstep = 1; P.qi0 = K.qbase + K.qlen - K.qs; } else { P.qstep = -1; P.qi0 = K.qbase + K.qs - 1; }
            P.tb_off = P.cig_off = 0; P.pad[0] = P.pad[1] = 0;
            out[np] = P;
        }
        ++np;
    }
    int lr = K.rs, lq = K.qs;
    for (int w0 = 0; w0 < K.cnt; w0 += 64) {
        const int i = w0 + lane;
        int cr = 0x7fffffff, cq = 0x7fffffff;
        if (i < K.cnt) { const uint64_t a = ca[i]; cr = A_G(a) - go + 1; cq = A_Q(a) + 1; }
        // cuts of this window, in order
        int ncut = 0; uint64_t live = ~0ULL;                 // lanes after the last cut
        for (;;) {
            const bool cond = i < K.cnt && (i == K.cnt - 1 || (cq - lq >= min_ksw_len && cr - lr >= min_ksw_len));
            const uint64_t m = __ballot(cond) & live;
            if (!m) break;
            const int f = __ffsll((unsigned long long)m) - 1;
            const int fr = __shfl(cr, f), fq = __shfl(cq, f);
            if (lane == ncut) { cut_r[lane] = fr; cut_q[lane] = fq; prev_r[lane] = lr; prev_q[lane] = lq; }
            lr = fr; lq = fq; ++ncut;
            live = f == 63 ? 0ULL : ~0ULL << (f + 1);
        }
        if (PASS && ncut) {
            __syncthreads();
            if (lane < ncut) {
                const int pr = prev_r[lane], pq = prev_q[lane];
                DpProb P; P.m = cut_q[lane] - pq; P.n = cut_r[lane] - pr; P.chain = c; P.kind = 0;
                const int W = P.m + P.n > ADAPT_MAX_STEPS ? d_fill_band_wide(P.m, P.n, bw, band_q4) : d_fill_band(P.m, P.n, bw, band_q4), dl = P.n - P.m;
                P.dlo = d_even_lo((dl < 0 ? dl : 0) - W); P.dhi = (dl > 0 ? dl : 0) + W;
                if (bw_long > bw && (dl > bw || -dl > bw)) { P.kind = 5; P.dlo = d_even_lo(-ext_band); P.dhi = ext_band; }      // long-gap fill: two bands of the extension width (spec 3.11)
                else if (P.dhi - P.dlo + 1 > DP_DMAX) P.kind = 3;
                P.tstep = 1; P.ti0 = K.tbase + pr; P.qcomp = (int8_t)K.rev;
                if (K.rev) { P.qstep = -1; P.qi0 = K.qbase + K.qlen - 1 - pq; } else { P.qstep = 1; P.qi0 = K.qbase + pq; }
                P.tb_off = P.cig_off = 0; P.pad[0] = P.pad[1] = 0;
                out[np + lane] = P;
            }
            __syncthreads();
        }
        np += ncut;
    }
    if (K.qe < K.qlen && K.re < K.tlen) {    // right extension
        if (PASS && lane == 0) {
            int rq = K.qlen - K.qe, rt = K.tlen - K.re;
            int mq = rq < ext_max ? rq : ext_max, mt = rt < mq + ext_band ? rt : mq + ext_band;
            DpProb P; P.m = mq; P.n = mt; P.dlo = d_even_lo(-ext_band); P.dhi = ext_band; P.kind = 2; P.chain = c;
            P.tstep = 1; P.ti0 = K.tbase + K.re; P.qcomp = (int8_t)K.rev;
            if (K.rev) { P.qstep = -1; P.qi0 = K.qbase + K.qlen - 1 - K.qe; } else { P.qstep = 1; P.qi0 = K.qbase + K.qe; }
            P.tb_off = P.cig_off = 0; P.pad[0] = P.pad[1] = 0;
            out[np] = P;
        }
        ++np;
    }
    if (!PASS && lane == 0) nprob[c] = np;
}

// DP classes.  0-4: LDS-state kernel (z-drop extensions, very wide fills), by band width;
// 5-9: register kernel for gap-fill problems: (lanes per problem, diagonal pairs per lane) =
// (32,1) (64,1) (64,2) (64,4) (64,8)  ->  bands up to 64 / 128 / 256 / 512 / 1024 diagonals.
#define DP_NCLS 25
// 10-17: packed-int16 register kernel d_dp_pkr<LPP, R> for short gap fills, by band width.  One lane per problem:
// class 17 D <= 16 (R = 4), classes 10..13 D = 17-20 / 21-24 / 25-28 / 29-32 (R = 5 / 6 / 7 / 8: exact ranges, so that only
// the last register of a lane can straddle the upper band edge); two lanes: classes 14..16 D <= 40 / 48 / 64 (R = 5 / 6 / 8).
// 18: z-drop extensions with D <= 64 (four lanes, R = 4)
// 19-21: wide fills in int16 while the scores fit (steps <= pk_wide_steps): D <= 256 / 512 / 1024, 1 / 2 / 4 waves per problem
// 22: 65..128 diagonals, four lanes x R = 8 (16 problems per wave): the class of the retried fills (wide band of a ~200-base
//     segment = 100-130 diagonals), which the margin rule of the band spec makes ~1 % of all fills
// 23, 24: z-drop extensions with D <= 128 / 256 (eight / sixteen lanes, R = 4): presets with ext_band > 31 (ngmlr-ont, round 5)
__device__ __forceinline__ int d_dp_class(int kind, int D, int steps, int pk_max_steps, int pk_ext_steps, int pk_wide_steps, int pk_wide_maxd = 1024, int pk_ext_maxd = 64)
{
    if ((kind == 1 || kind == 2) && steps <= pk_ext_steps) {
        if (D <= 64) return 18;
        if (D <= 128 && pk_ext_maxd >= 128) return 23;
        if (D <= 256 && pk_ext_maxd >= 256) return 24;
    }
    if (kind == 0 && steps <= pk_max_steps) {
        if (D <= 16) return 17;
        if (D <= 20) return 10;
        if (D <= 24) return 11;
        if (D <= 28) return 12;
        if (D <= 32) return 13;
        if (D <= 40) return 14;
        if (D <= 48) return 15;
        if (D <= 64) return 16;
        if (D <= 128) return 22;
    }
    if (kind == 0 && steps <= pk_wide_steps && D > 64 && D <= pk_wide_maxd) {
        if (D <= 256) return 19;
        if (D <= 512) return 20;
        if (D <= 1024) return 21;
    }
    if (kind == 0) {
        if (D <= 64) return 5;
        if (D <= 128) return 6;
        if (D <= 256) return 7;
        if (D <= 512) return 8;
        if (D <= 1024) return 9;
    }
    return D <= 64 ? 0 : D <= 128 ? 1 : D <= 256 ? 2 : D <= 1024 ? 3 : 4;
}
// Four bits per cell.  The one-piece cell (classes 17 and 10 when the preset's gap costs allow it: d_cell_pk) has three
// sources and two extension flags -- a nibble -- once the "bases equal" bit is gone, and that bit is not needed: with one
// affine piece and no ambiguous base the number of matching columns follows from the score of the path,
// a * match - b * (M columns - match) - sum over gap runs (q + e * length) = score.  Row k (anti-diagonals 2k, 2k+1) of
// register r is one 16-bit word {even step: low nibbles, odd step: high nibbles} x {low-half diagonal, high-half
// diagonal}; a row is ceil(R/2) dwords, a row PAIR a whole number of 8-byte units of the wave-interleaved layout.
__host__ __device__ __forceinline__ int d_onep_d(int q, int e, int q2, int e2) { return e > e2 ? (q2 - q + (e - e2) - 1) / (e - e2) : 1 << 20; }
__host__ __device__ __forceinline__ bool d_tb4(int cls, int tb4) { return (cls == 17 && (tb4 & 1)) || (cls == 10 && (tb4 & 2)); }
__host__ __device__ __forceinline__ int d_tb4_rowb(int cls) { return cls == 17 ? 8 : 12; }       // bytes per row: 4 * ceil(R / 2)
// the z-drop extension classes spill in tiles of four rows (d_dp_pkx): a 64-byte line = four consecutive rows of 16 bytes
__host__ __device__ __forceinline__ bool d_tb_tiled(int cls) { return cls == 18 || cls == 23 || cls == 24; }
// dwords per packed trace-back row
__device__ __forceinline__ int d_cls_slots(int cls)
{
    if (cls == 22 || cls == 23) return 32;
    if (cls == 24) return 64;
    if (cls >= 19) return 64 << (cls - 19);
    if (cls >= 10) return cls <= 13 ? cls - 5 : cls == 14 ? 10 : cls == 15 ? 12 : cls == 17 ? 4 : 16;
    return cls == 5 ? 32 : 64 << (cls - 6);
}
// any ambiguous base among the n bases from absolute index lo on?
__device__ __forceinline__ bool d_any_n(const uint32_t *__restrict__ nmask, int64_t lo, int n)
{
    if (n <= 0) return false;
    const int64_t hi = lo + n - 1;
    for (int64_t w = lo >> 5; w <= (hi >> 5); ++w) {
        uint32_t m = nmask[w];
        if (w == (lo >> 5)) m &= ~0u << (int)(lo & 31);
        if (w == (hi >> 5)) m &= ~0u >> (31 - (int)(hi & 31));
        if (m) return true;
    }
    return false;
}
__global__ void k_prob_sizes(DpProb *__restrict__ probs, int32_t np, int fill_margin, int32_t pk_max_steps, int32_t pk_ext_steps, int32_t pk_wide_steps,
                             const uint32_t *__restrict__ qnmask, const uint32_t *__restrict__ tnmask,
                             int64_t *__restrict__ tb_bytes, int64_t *__restrict__ cig_ops, int32_t tb4, int32_t tb4_steps, int32_t pk_wide_maxd, int32_t pk_ext_maxd)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    const DpProb P = probs[i];
    int D = P.dhi - P.dlo + 1, stride = (D + 2) / 2;
    int cls = d_dp_class(P.kind, D, P.m + P.n, pk_max_steps, pk_ext_steps, pk_wide_steps, pk_wide_maxd, pk_ext_maxd);
    if (cls >= 10 && P.kind < 3) {
        // the packed kernels have no ambiguity case: a problem with an N inside either window takes an int32 class
        const bool hasn = d_any_n(qnmask, P.qstep > 0 ? P.qi0 : P.qi0 - P.m + 1, P.m) || d_any_n(tnmask, P.tstep > 0 ? P.ti0 : P.ti0 - P.n + 1, P.n);
        if (hasn) cls = d_dp_class(P.kind, D, P.m + P.n, 0, 0, 0);
    }
    if (d_tb4(cls, tb4) && P.m + P.n > tb4_steps) cls = 14;        // the nibble cell keeps scores times four: longer fills take the two-lane class
    int64_t tb;
    if (P.kind == 5) {       // both halves: H of every band cell (int32) + its trace-back byte, anti-diagonal major
        const int S = P.m < P.n ? P.m : P.n, lm = P.m < S + P.dhi ? P.m : S + P.dhi, ln = P.n < S + P.dhi ? P.n : S + P.dhi;
        tb = (2 * (int64_t)(lm + ln + 1) * (D * 4 + stride) + 127) & ~127LL;
    }
    else if (P.kind >= 3) tb = 0;
    else if (d_tb4(cls, tb4)) tb = ((int64_t)(((P.m + P.n) / 2 + 2) / 2) * (2 * d_tb4_rowb(cls)) + 63) & ~63LL;     // row pairs of nibbles
    else if (d_tb_tiled(cls)) tb = (int64_t)(((P.m + P.n) / 2 + 4) / 4) * d_cls_slots(cls) * 16;     // tiles of four rows (d_dp_pkx)
    else if (cls >= 10) tb = ((int64_t)((P.m + P.n) / 2 + 1) * d_cls_slots(cls) * 4 + 63) & ~63LL;     // whole 64-byte lines (d_traceback_rows)
    else if (cls >= 5) tb = (int64_t)((P.m + P.n) / 4 + 1) * d_cls_slots(cls) * 4;
    else tb = ((int64_t)(P.m + P.n + 1) * stride + 127) & ~127LL;
    int cells = 0;
    if (cls >= 10 && cls != 18 && cls < 23) for (int d = P.dlo; d <= P.dhi; ++d) {
        int ilo = d < 0 ? 1 - d : 1, ihi = P.n - d < P.m ? P.n - d : P.m;
        if (ihi >= ilo) cells += ihi - ilo + 1;
    }
    probs[i].pad[0] = cls | (fill_margin << 8); probs[i].pad[1] = cells;      // class, and the margin of the retry test for the trace-back
    tb_bytes[i] = tb;
    if (P.kind == 5) { const int S = P.m < P.n ? P.m : P.n; cig_ops[i] = 3 * (int64_t)(2 * (S + P.dhi) + 2); }      // final ops + room for the right half's walk
    else cig_ops[i] = P.kind == 3 ? 2 : (int64_t)P.m + P.n;
}
// scratch offsets of every problem, class histogram, and the (key, problem) pairs whose ONE radix sort yields all class
// lists at once: key = class << 19 | (0x7FFFF - steps), so a class is a contiguous range ordered by decreasing steps
struct ClsOff { int32_t off[DP_NCLS + 1]; };
// Trace-back offsets in CLASS-LIST order: the 64 problems of a wave (neighbours in their class list) get adjacent pieces of
// the trace-back buffer.  With offsets in problem order every lane of a wave worked in a different page of a 10-GB buffer and
// both the spill and the walk ran at the rate of the address translation (tools/ubench/tb_pattern.hip: scattered 64-byte
// reads 2.6 TB/s inside 1 GiB, 1.0 TB/s over 4 GiB or more).
// The one-lane-per-problem classes (10-13, 17) go one step further: the 64 problems of a wave are INTERLEAVED in units of
// 8 bytes (unit u of lane t at wave base + (u*64 + t)*8), so every store instruction of the forward pass and every load
// instruction of the walk covers 512 contiguous bytes and a whole row pair / line of the wave is one contiguous 2.5-4 KB
// block: both kernels stream instead of touching 64 different lines per instruction.  The wave's piece is sized by its
// first (longest) problem; tb_off of a problem points at its unit 0.
__device__ __forceinline__ bool d_tb_interleaved(int cls) { return (cls >= 10 && cls <= 13) || cls == 17; }
__device__ __forceinline__ int d_cls_of_pos(const ClsOff &off, int i) { int c = 0; while (c < DP_NCLS - 1 && i >= off.off[c + 1]) ++c; return c; }
__global__ void __launch_bounds__(256) k_tb_gather(const int64_t *__restrict__ tb_bytes, const int32_t *__restrict__ list, int32_t np, ClsOff off, int64_t *__restrict__ out)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == np) out[np] = 0;
    if (i >= np) return;
    const int c = d_cls_of_pos(off, i);
    if (!d_tb_interleaved(c)) { out[i] = tb_bytes[list[i]]; return; }
    const int first = off.off[c] + ((i - off.off[c]) & ~63);
    const int last = first + 63 < off.off[c + 1] - 1 ? first + 63 : off.off[c + 1] - 1;
    out[i] = i == last ? tb_bytes[list[first]] * 64 : 0;           // the whole wave's piece, counted once (at its last lane)
}
__global__ void __launch_bounds__(256) k_tb_scatter(DpProb *__restrict__ probs, const int32_t *__restrict__ list, int32_t np, ClsOff off, const int64_t *__restrict__ toff)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    const int c = d_cls_of_pos(off, i);
    probs[list[i]].tb_off = toff[i] + (d_tb_interleaved(c) ? (int64_t)((i - off.off[c]) & 63) * 8 : 0);
}
__global__ void __launch_bounds__(256) k_prob_assign(DpProb *__restrict__ probs, int32_t np, const int64_t *__restrict__ tb_off, const int64_t *__restrict__ cig_off,
                                                     int32_t *__restrict__ cls_cnt, uint32_t *__restrict__ sort_key, int32_t *__restrict__ sort_val)
{
    __shared__ int32_t lcnt[DP_NCLS];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < DP_NCLS) lcnt[threadIdx.x] = 0;
    __syncthreads();
    if (i < np) {
        probs[i].tb_off = tb_off[i];
        if (cig_off) probs[i].cig_off = cig_off[i];       // a retry keeps the CIGAR slot of the original problem
        const int c = probs[i].kind >= 3 ? 0 : (probs[i].pad[0] & 0xff);
        atomicAdd(&lcnt[c], 1);
        int steps = probs[i].m + probs[i].n; if (steps > 0x7FFFF) steps = 0x7FFFF;
        sort_key[i] = (uint32_t)c << 19 | (uint32_t)(0x7FFFF - steps);          // 24 bits: three radix passes instead of four
        sort_val[i] = i;
    }
    __syncthreads();
    if (threadIdx.x < DP_NCLS && lcnt[threadIdx.x]) atomicAdd(&cls_cnt[threadIdx.x], lcnt[threadIdx.x]);
}

// retry pass: problems whose narrow-band path touched a band edge are re-aligned with the wide band
__global__ void k_retry_collect(const int32_t *__restrict__ flag, int32_t np, int32_t *__restrict__ cnt, int32_t *__restrict__ list)
{
    // one atomic per wave (ballot + rank), not one per retried problem: tens of thousands of additions to ONE address
    // serialise in L2 (measured: up to 9 ms for 61 k retries)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool f = i < np && flag[i];
    const uint64_t m = __ballot(f);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == __ffsll((unsigned long long)m) - 1) base = atomicAdd(cnt, __popcll(m));
    base = __shfl(base, __ffsll((unsigned long long)m) - 1);
    if (f) list[base + __popcll(m & ((1ULL << lane) - 1))] = i;
}
__global__ void k_retry_build(const DpProb *__restrict__ probs, const int32_t *__restrict__ list, int32_t n, int32_t bw, int32_t band_q4, DpProb *__restrict__ out)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    DpProb P = probs[list[k]];
    const int W = d_fill_band(P.m, P.n, bw, band_q4), W2 = d_fill_band_wide(P.m, P.n, bw, band_q4), dl = P.n - P.m;
    const int lo = d_even_lo((dl < 0 ? dl : 0) - W2), hi = (dl > 0 ? dl : 0) + W2;
    P.chain = list[k];                  // index of the original problem
    P.pad[1] = 0;
    if (W2 > W && hi - lo + 1 <= DP_DMAX) { P.dlo = lo; P.dhi = hi; P.kind = 0; } else P.kind = 4;   // 4 = nothing to redo
    out[k] = P;
}
__global__ void k_retry_merge(const DpProb *__restrict__ probs2, const DpRes *__restrict__ res2, int32_t n, DpRes *__restrict__ res)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n || probs2[k].kind == 4) return;
    DpRes r = res2[k];
    r.cells += res[probs2[k].chain].cells;
    res[probs2[k].chain] = r;
}

struct DpArgs {
    const uint32_t *qseq2, *qnmask, *tseq2, *tnmask;
    int64_t qtot, ttot;      // padded base counts of the two packed arrays
    const DpProb *probs; const int32_t *list; int32_t nlist;
    DpOpt o;
    uint8_t *tb; uint32_t *cig; DpRes *res;
    int32_t dcap;            // diagonals of LDS state per wave (LDS kernel)
    int32_t *retry;
    int32_t tb4;             // bit 0: class 17, bit 1: class 10 spill 4 bits per cell (one-piece cell only; d_tb4)
    int32_t tag8_steps;      // a wave of the two-piece classes whose longest problem has at most this many steps takes d_cell_pk8 (0 = never)
};

__device__ __forceinline__ int64_t d_wave_max64(int64_t v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        int64_t w = (int64_t)((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)v, o) | (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)((uint64_t)v >> 32), o) << 32);
        v = w > v ? w : v;
    }
    return v;
}

// one DP cell (two-piece affine); returns the trace-back byte.  left = (i,j-1), up = (i-1,j), hd = (i-1,j-1)
__device__ __forceinline__ uint32_t d_cell(const DpOpt &o, int32_t hd, int32_t hl, int32_t e1l, int32_t e2l, int32_t hu, int32_t f1u, int32_t f2u,
                                           int qb, int tbv, int32_t &h, int32_t &ve1, int32_t &vf1, int32_t &ve2, int32_t &vf2)
{
    uint32_t t = 0; int32_t op, g;
    if (o.cx_scale) {        // convex cost: e2l / f2u are the lengths of the gaps E(left) / F(up) end with; ve2 / vf2 the new lengths
        const int le = e2l < 0 ? 0 : e2l > o.cx_flat ? o.cx_flat : e2l, lf = f2u < 0 ? 0 : f2u > o.cx_flat ? o.cx_flat : f2u;
        op = hl - o.cx_open - d_cx_ext(o, 0); g = e1l - d_cx_ext(o, le); if (g > op) { ve1 = g; t |= 8; ve2 = le + 1; } else { ve1 = op; ve2 = 1; }
        op = hu - o.cx_open - d_cx_ext(o, 0); g = f1u - d_cx_ext(o, lf); if (g > op) { vf1 = g; t |= 16; vf2 = lf + 1; } else { vf1 = op; vf2 = 1; }
        if (ve2 > o.cx_flat) ve2 = o.cx_flat;
        if (vf2 > o.cx_flat) vf2 = o.cx_flat;
        int scx;
        if (qb > 3 || tbv > 3) scx = -o.sc_ambi; else if (qb == tbv) { scx = o.a; t |= 128; } else scx = -o.b;
        h = hd + scx; uint32_t srcx = 0;
        if (ve1 > h) { h = ve1; srcx = 1; }
        if (vf1 > h) { h = vf1; srcx = 2; }
        return t | srcx;
    }
    op = hl - o.q - o.e;   g = e1l - o.e;  if (g > op) { ve1 = g; t |= 8; }  else ve1 = op;
    op = hu - o.q - o.e;   g = f1u - o.e;  if (g > op) { vf1 = g; t |= 16; } else vf1 = op;
    op = hl - o.q2 - o.e2; g = e2l - o.e2; if (g > op) { ve2 = g; t |= 32; } else ve2 = op;
    op = hu - o.q2 - o.e2; g = f2u - o.e2; if (g > op) { vf2 = g; t |= 64; } else vf2 = op;
    int sc;
    if (qb > 3 || tbv > 3) sc = -o.sc_ambi; else if (qb == tbv) { sc = o.a; t |= 128; } else sc = -o.b;
    h = hd + sc; uint32_t src = 0;
    if (ve1 > h) { h = ve1; src = 1; }
    if (vf1 > h) { h = vf1; src = 2; }
    if (ve2 > h) { h = ve2; src = 3; }
    if (vf2 > h) { h = vf2; src = 4; }
    return t | src;
}

// ---- long-gap fill (spec 3.11; oracle longgap_fill): a segment whose lengths differ by more than bw.  One wave.
// LEFT = global banded DP from the start in |j - i| <= W over the first min(len, S + W) rows / columns, RIGHT = the same from the
// end on the reversed sequences; H of every band cell and its trace-back byte go to the problem's scratch; the junction is the
// short-axis coordinate that maximises  max_k [HL + e2 k] + max_k [HR + e2 k]  (smallest coordinate, then smallest k on ties);
// lane 0 walks both halves and writes the run-merged ops end -> start: RIGHT reversed, the gap, LEFT.
// d_longgap is single-wave code: its steps are ordered by a WAVE-level barrier (LDS operations of one wave are executed in order; the
// fence keeps the compiler from moving them across).  Until round 5 it called __syncthreads(), also from k_dp_w4, whose waves 1-3 had
// returned by then -- a barrier that not every thread of the block reaches is undefined in the HIP model.
__device__ __forceinline__ void d_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }      // (workgroup scope: the halves' H / trace-back bytes go through global memory)
__device__ __forceinline__ void d_longgap_half(const DpArgs &A, const DpProb &P, int side, int lm, int ln, int m, int n, int lane, int32_t *lds,
                                               int32_t *Hg, uint8_t *tbg, int &ncell)
{
    const DpOpt o = A.o;
    const int dlo = P.dlo, dhi = P.dhi, D = dhi - dlo + 1, stride = (D + 2) / 2;
    const int qstep = side ? -P.qstep : P.qstep, tstep = side ? -P.tstep : P.tstep;
    const int64_t qi0 = side ? P.qi0 + (int64_t)P.qstep * (m - 1) : P.qi0, ti0 = side ? P.ti0 + (int64_t)P.tstep * (n - 1) : P.ti0;
    int32_t *H = lds, *E1 = H + (A.dcap + 2), *F1 = E1 + (A.dcap + 2), *E2 = F1 + (A.dcap + 2), *F2 = E2 + (A.dcap + 2);
    d_wave_sync();
    for (int x = lane; x < D + 2; x += 64) { H[x] = TELR_NEG; E1[x] = TELR_NEG; F1[x] = TELR_NEG; E2[x] = TELR_NEG; F2[x] = TELR_NEG; }
    d_wave_sync();
    if (lane == 0) { H[0 - dlo + 1] = 0; Hg[0 - dlo] = 0; }
    d_wave_sync();
    for (int a = 1; a <= lm + ln; ++a) {
        int d0 = -a > dlo ? -a : dlo; if (a - 2 * lm > d0) d0 = a - 2 * lm;
        int d1 = a < dhi ? a : dhi;   if (2 * ln - a < d1) d1 = 2 * ln - a;
        if (((d0 - a) & 1) != 0) ++d0;
        for (int d = d0 + 2 * lane; d <= d1; d += 128) {
            const int i = (a - d) >> 1, j = (a + d) >> 1, x = d - dlo + 1;
            int32_t h, ve1, vf1, ve2, vf2;
            if (i == 0) { ve1 = -(o.q + j * o.e); ve2 = -(o.q2 + j * o.e2); vf1 = vf2 = TELR_NEG; h = ve1 > ve2 ? ve1 : ve2; }
            else if (j == 0) { vf1 = -(o.q + i * o.e); vf2 = -(o.q2 + i * o.e2); ve1 = ve2 = TELR_NEG; h = vf1 > vf2 ? vf1 : vf2; }
            else {
                int qb = d_base(A.qseq2, A.qnmask, qi0 + (int64_t)qstep * (i - 1));
                int tbv = d_base(A.tseq2, A.tnmask, ti0 + (int64_t)tstep * (j - 1));
                if (P.qcomp && qb < 4) qb = 3 - qb;
                const uint32_t t = d_cell(o, H[x], H[x - 1], E1[x - 1], E2[x - 1], H[x + 1], F1[x + 1], F2[x + 1], qb, tbv, h, ve1, vf1, ve2, vf2);
                tbg[(int64_t)a * stride + ((d - dlo) >> 1)] = (uint8_t)t;
                ++ncell;
            }
            if (h < TELR_NEG) h = TELR_NEG;
            if (ve1 < TELR_NEG) ve1 = TELR_NEG;
            if (vf1 < TELR_NEG) vf1 = TELR_NEG;
            if (ve2 < TELR_NEG) ve2 = TELR_NEG;
            if (vf2 < TELR_NEG) vf2 = TELR_NEG;
            H[x] = h; E1[x] = ve1; F1[x] = vf1; E2[x] = ve2; F2[x] = vf2;
            Hg[(int64_t)a * D + (d - dlo)] = h;
        }
        d_wave_sync();
    }
}
// walk from (i, j) to (0, 0) through a half's trace-back bytes: run-merged ops (end -> start) appended at out[no...]; counts M columns and equal bases
__device__ __forceinline__ int d_longgap_walk(const uint8_t *tbg, int dlo, int stride, int i, int j, uint32_t *out, int no, int &mcols, int &mlen)
{
    int state = 0;
    auto push = [&](uint32_t op) { if (no > 0 && (out[no - 1] & 0xfu) == op) out[no - 1] += 16u; else out[no++] = 16u | op; };
    while (i > 0 && j > 0) {
        const uint32_t t = tbg[(int64_t)(i + j) * stride + ((j - i - dlo) >> 1)];
        if (state == 0) state = (int)(t & 7u);
        if (state == 0) { push(0u); ++mcols; mlen += (int)(t >> 7 & 1u); --i; --j; }
        else if (state == 1) { push(2u); if (!(t & 8u))  state = 0; --j; }
        else if (state == 2) { push(1u); if (!(t & 16u)) state = 0; --i; }
        else if (state == 3) { push(2u); if (!(t & 32u)) state = 0; --j; }
        else                 { push(1u); if (!(t & 64u)) state = 0; --i; }
    }
    if (i > 0) { if (no > 0 && (out[no - 1] & 0xfu) == 1u) out[no - 1] += (uint32_t)i << 4; else out[no++] = (uint32_t)i << 4 | 1u; }
    if (j > 0) { if (no > 0 && (out[no - 1] & 0xfu) == 2u) out[no - 1] += (uint32_t)j << 4; else out[no++] = (uint32_t)j << 4 | 2u; }
    return no;
}
__device__ void d_longgap(const DpArgs &A, const DpProb &P, int prob, int lane, int32_t *lds)
{
    const DpOpt o = A.o;
    const int m = P.m, n = P.n, ins = m > n, S = ins ? n : m, W = P.dhi, dlo = P.dlo, D = P.dhi - dlo + 1, stride = (D + 2) / 2;
    const int lm = m < S + W ? m : S + W, ln = n < S + W ? n : S + W, na = lm + ln + 1;
    uint8_t *base = A.tb + P.tb_off;
    int32_t *HL = (int32_t*)base, *HR = HL + (int64_t)na * D;
    uint8_t *tbL = (uint8_t*)(HR + (int64_t)na * D), *tbR = tbL + (int64_t)na * stride;
    int ncell = 0;
    d_longgap_half(A, P, 0, lm, ln, m, n, lane, lds, HL, tbL, ncell);
    d_longgap_half(A, P, 1, lm, ln, m, n, lane, lds, HR, tbR, ncell);
    __threadfence_block();
    d_wave_sync();
    // ---- the junction
    const int lim = ins ? lm : ln;
    int64_t bestk = INT64_MIN; int b_al = 0, b_ar = 0;
    for (int c = lane; c <= S; c += 64) {
        int64_t vl = INT64_MIN, vr = INT64_MIN; int al = 0, ar = 0;
        for (int k = c - W; k <= c + W; ++k) {
            if (k < 0 || k > lim) continue;
            const int i = ins ? k : c, j = ins ? c : k, d = j - i;
            if (d < dlo || d > P.dhi) continue;
            const int32_t h = HL[(int64_t)(i + j) * D + (d - dlo)];
            if (h <= TELR_NEG / 2) continue;
            const int64_t v = (int64_t)h + (int64_t)o.e2 * k;
            if (v > vl) { vl = v; al = k; }
        }
        const int c2 = S - c;
        for (int k = c2 - W; k <= c2 + W; ++k) {
            if (k < 0 || k > lim) continue;
            const int i = ins ? k : c2, j = ins ? c2 : k, d = j - i;
            if (d < dlo || d > P.dhi) continue;
            const int32_t h = HR[(int64_t)(i + j) * D + (d - dlo)];
            if (h <= TELR_NEG / 2) continue;
            const int64_t v = (int64_t)h + (int64_t)o.e2 * k;
            if (v > vr) { vr = v; ar = k; }
        }
        if (vl == INT64_MIN || vr == INT64_MIN) continue;
        if ((ins ? m : n) - al - ar < 1) continue;
        const int64_t key = (vl + vr) * 65536LL + (int64_t)(0xffff - c);          // largest total, smallest c (lanes visit their c ascending; S < 65536)
        if (key > bestk) { bestk = key; b_al = al; b_ar = ar; }
    }
    const int64_t wk = d_wave_max64(bestk);
    const uint64_t who = __ballot(bestk == wk);
    const int src = __ffsll((unsigned long long)who) - 1;
    const int bc = 0xffff - (int)(wk & 0xffff), al = __shfl(b_al, src), ar = __shfl(b_ar, src);
    int nc = ncell;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) nc += __shfl_xor(nc, s);
    if (lane == 0) {
        const int g = (ins ? m : n) - al - ar;
        const int li = ins ? al : bc, lj = ins ? bc : al, ri = ins ? ar : S - bc, rj = ins ? S - bc : ar;
        const int c1 = o.q + g * o.e, c2g = o.q2 + g * o.e2;
        DpRes R; R.bi = m; R.bj = n; R.tbases = n; R.cells = nc;
        R.score = HL[(int64_t)(li + lj) * D + (lj - li - dlo)] + HR[(int64_t)(ri + rj) * D + (rj - ri - dlo)] - (c1 < c2g ? c1 : c2g);
        uint32_t *out = A.cig + P.cig_off, *tmp = out + 2 * (2 * (S + W) + 2);
        int mcols = 0, mlen = 0;
        // RIGHT's walk runs junction -> end in the original orientation: taken into tmp, then emitted backwards
        const int nr = d_longgap_walk(tbR, dlo, stride, ri, rj, tmp, 0, mcols, mlen);
        int no = 0;
        for (int z = nr - 1; z >= 0; --z) { if (no > 0 && (out[no - 1] & 0xfu) == (tmp[z] & 0xfu)) out[no - 1] += tmp[z] & ~0xfu; else out[no++] = tmp[z]; }
        { const uint32_t gop = ins ? 1u : 2u; if (no > 0 && (out[no - 1] & 0xfu) == gop) out[no - 1] += (uint32_t)g << 4; else out[no++] = (uint32_t)g << 4 | gop; }
        no = d_longgap_walk(tbL, dlo, stride, li, lj, out, no, mcols, mlen);
        R.nops = no; R.mlen = mlen; R.mcols = mcols;
        A.res[prob] = R;
    }
}

// ---- LDS-state forward kernel: any band width up to DP_DMAX, fills and z-drop extensions.
// One wave per problem; DP state per diagonal in LDS, updated in place (cells of one
// anti-diagonal touch only the other parity's diagonals).  Trace-back bytes go to
// tb[a][slot]; the walk itself is done by k_traceback.
// NT = 64: one wave per problem (every class).  NT = 256 (round 5, classes 3 and 4: bands of up to DP_DMAX diagonals): four waves sweep an
// anti-diagonal together -- the polishing map's `-r2k` fills (a few thousand problems of up to 4,000 diagonals, 760 Mcells per 1,000 loci) were
// single waves of 30 ms; only fills go there (z-drop extensions are at most 255 diagonals wide: classes 0-2), the long-gap and fallback
// kinds are done by the block's first wave.
template <int NT> __device__ __forceinline__ void d_dp_lds(const DpArgs &A)
{
    extern __shared__ __align__(16) int32_t lds[];
    const int pi = blockIdx.x;
    if (pi >= A.nlist) return;
    const int prob = A.list[pi];
    const DpProb P = A.probs[prob];
    const int lane = threadIdx.x;
    if (P.kind == 4) return;
    __builtin_amdgcn_s_setprio(3);      // latency-bound single wave: do not wait behind the bulk kernels' waves
    const int m = P.m, n = P.n, dlo = P.dlo, dhi = P.dhi, D = dhi - dlo + 1, stride = (D + 2) / 2;
    const DpOpt o = A.o;
    const bool ext = P.kind == 1 || P.kind == 2;
    DpRes R; R.score = 0; R.bi = 0; R.bj = 0; R.nops = 0; R.mlen = 0; R.cells = 0; R.tbases = n; R.mcols = 0;
    int ncell = 0;

    if (NT > 64 && (P.kind == 5 || P.kind == 3) && lane >= 64) return;      // (single-wave code below, without workgroup barriers: d_wave_sync)
    if (P.kind == 5) { d_longgap(A, P, prob, lane, lds); return; }
    if (P.kind == 3) {
        // band wider than the engine accepts: diagonal + one closing gap (oracle band_dp_fallback)
        int mn = m < n ? m : n, sc = 0, ml = 0;
        for (int x = lane; x < mn; x += 64) {
            int qb = d_base(A.qseq2, A.qnmask, P.qi0 + (int64_t)P.qstep * x), tbv = d_base(A.tseq2, A.tnmask, P.ti0 + (int64_t)P.tstep * x);
            if (P.qcomp && qb < 4) qb = 3 - qb;
            if (qb > 3 || tbv > 3) sc -= o.sc_ambi; else if (qb == tbv) { sc += o.a; ++ml; } else sc -= o.b;
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) { sc += __shfl_xor(sc, s); ml += __shfl_xor(ml, s); }
        int g = m > n ? m - n : n - m;
        if (g) { if (o.cx_scale) sc -= d_cx_cost(o, g); else { int c1 = o.q + g * o.e, c2 = o.q2 + g * o.e2; sc -= c1 < c2 ? c1 : c2; } }
        if (lane == 0) {
            int no = 0;
            if (g) A.cig[P.cig_off + no++] = (uint32_t)g << 4 | (m > n ? 1u : 2u);
            if (mn) A.cig[P.cig_off + no++] = (uint32_t)mn << 4;
            R.score = sc; R.bi = m; R.bj = n; R.nops = no; R.mlen = ml; R.mcols = mn;
            A.res[prob] = R;
        }
        return;
    }

    int32_t *H = lds, *E1 = H + (A.dcap + 2), *F1 = E1 + (A.dcap + 2), *E2 = F1 + (A.dcap + 2), *F2 = E2 + (A.dcap + 2);
    for (int x = lane; x < D + 2; x += NT) { H[x] = TELR_NEG; E1[x] = TELR_NEG; F1[x] = TELR_NEG; E2[x] = TELR_NEG; F2[x] = TELR_NEG; }
    __syncthreads();
    if (lane == 0 && 0 >= dlo && 0 <= dhi) H[0 - dlo + 1] = 0;
    __syncthreads();
    uint8_t *tb = A.tb + P.tb_off;
    int best = 0, bi = 0, bj = 0, prev_cur = TELR_NEG;
    for (int a = 1; a <= m + n; ++a) {
        int d0 = -a > dlo ? -a : dlo; if (a - 2 * m > d0) d0 = a - 2 * m;
        int d1 = a < dhi ? a : dhi;   if (2 * n - a < d1) d1 = 2 * n - a;
        if (((d0 - a) & 1) != 0) ++d0;
        int64_t curk = INT64_MIN;
        for (int d = d0 + 2 * lane; d <= d1; d += 2 * NT) {
            const int i = (a - d) >> 1, j = (a + d) >> 1, x = d - dlo + 1;
            int32_t h, ve1, vf1, ve2, vf2;
            if (i == 0 && o.cx_scale) { ve1 = -d_cx_cost(o, j); ve2 = j < o.cx_flat ? j : o.cx_flat; vf1 = TELR_NEG; vf2 = 0; h = ve1; }
            else if (j == 0 && o.cx_scale) { vf1 = -d_cx_cost(o, i); vf2 = i < o.cx_flat ? i : o.cx_flat; ve1 = TELR_NEG; ve2 = 0; h = vf1; }
            else if (i == 0) {
                ve1 = -(o.q + j * o.e); ve2 = -(o.q2 + j * o.e2); vf1 = vf2 = TELR_NEG;
                h = ve1 > ve2 ? ve1 : ve2;
            } else if (j == 0) {
                vf1 = -(o.q + i * o.e); vf2 = -(o.q2 + i * o.e2); ve1 = ve2 = TELR_NEG;
                h = vf1 > vf2 ? vf1 : vf2;
            } else {
                int qb = d_base(A.qseq2, A.qnmask, P.qi0 + (int64_t)P.qstep * (i - 1));
                int tbv = d_base(A.tseq2, A.tnmask, P.ti0 + (int64_t)P.tstep * (j - 1));
                if (P.qcomp && qb < 4) qb = 3 - qb;
                uint32_t t = d_cell(o, H[x], H[x - 1], E1[x - 1], E2[x - 1], H[x + 1], F1[x + 1], F2[x + 1], qb, tbv, h, ve1, vf1, ve2, vf2);
                tb[(int64_t)a * stride + ((d - dlo) >> 1)] = (uint8_t)t;
                ++ncell;
            }
            if (h < TELR_NEG) h = TELR_NEG;
            if (ve1 < TELR_NEG) ve1 = TELR_NEG;
            if (vf1 < TELR_NEG) vf1 = TELR_NEG;
            if (ve2 < TELR_NEG) ve2 = TELR_NEG;
            if (vf2 < TELR_NEG) vf2 = TELR_NEG;
            H[x] = h; E1[x] = ve1; F1[x] = vf1; E2[x] = ve2; F2[x] = vf2;
            int64_t kk = (int64_t)h * 4294967296LL + (int64_t)(0x7fffffff - (d - dlo));   // max h, smallest d among ties
            curk = kk > curk ? kk : curk;
        }
        __syncthreads();   // orders this step's LDS writes before the next step's reads
        if (ext) {
            curk = d_wave_max64(curk);
            if (NT > 64) {                   // (no extension is this wide today -- ext_band <= 127 -- but the step's maximum is the block's)
                __shared__ long long wk[NT / 64];
                if ((lane & 63) == 0) wk[lane >> 6] = curk;
                __syncthreads();
                for (int w = 0; w < NT / 64; ++w) curk = wk[w] > curk ? wk[w] : curk;
                __syncthreads();
            }
            int cur = TELR_NEG, cur_d = 0;
            if (curk != INT64_MIN) { cur = (int)(curk >> 32); cur_d = 0x7fffffff - (int)(curk & 0xffffffffLL) + dlo; }
            if (cur > best) { best = cur; bi = (a - cur_d) >> 1; bj = (a + cur_d) >> 1; }
            int c2 = cur > prev_cur ? cur : prev_cur;
            if (best - c2 > o.zdrop) break;
            prev_cur = cur;
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) ncell += __shfl_xor(ncell, s);
    if (NT > 64) {                       // the waves' cell counts through LDS (the DP state is done with)
        __shared__ int32_t wcells[NT / 64];
        if ((lane & 63) == 0) wcells[lane >> 6] = ncell;
        __syncthreads();
        ncell = 0;
        for (int w = 0; w < NT / 64; ++w) ncell += wcells[w];
    }
    if (lane == 0) {
        if (ext) { R.score = best; R.bi = bi; R.bj = bj; }
        else { R.score = H[(n - m) - dlo + 1]; R.bi = m; R.bj = n; }
        R.cells = ncell;
        A.res[prob] = R;
    }
}
__global__ void __launch_bounds__(64) k_dp(DpArgs A) { d_dp_lds<64>(A); }
__global__ void __launch_bounds__(256) k_dp_w4(DpArgs A) { d_dp_lds<256>(A); }

// ---- register forward kernel for gap-fill problems (global alignment, D <= 2*R*LPP).
// LPP lanes per problem (64/LPP problems per wave); lane l owns the 2R consecutive
// diagonals dlo+2R*l .. dlo+2R*l+2R-1 in registers (dlo is even), as R (even, odd) pairs.
// Odd anti-diagonals update the odd diagonals, even ones the even diagonals, so every lane
// computes R cells per step; the only values that cross lanes are the (H, F1, F2) of the
// next lane's first even diagonal (odd steps) or the (H, E1, E2) of the previous lane's last
// odd diagonal (even steps): three DPP wave shifts per step, no LDS.
//   * boundary cells (row 0 / column 0) are pre-loaded into the registers, so the loop has no
//     boundary branches; each diagonal's interior cells are live for a in [alo, alo+span];
//   * bases stream through per-lane shift registers (32 bases per 64-bit refill, refilled
//     on a wave-uniform cadence): one new query base per even step, one new target base per
//     odd step, handed from pair to pair inside the lane;
//   * trace-back bytes are packed four steps per dword: tb32[(a>>2)*SLOTS + l*R + r], byte a&3.
// Unreachable cells are not clamped to NEG (the oracle clamps): such values are never
// selected by a reachable cell, so scores and CIGARs are identical.
struct BaseStream { uint64_t w; uint32_t nm; };

// 32 bases starting at absolute base index `start` (may lie outside the array), base b at bits 2*(b-start)
__device__ __forceinline__ void d_load32(const uint32_t *__restrict__ seq2, const uint32_t *__restrict__ nmask, int64_t start, int64_t tot,
                                         uint64_t &w, uint32_t &nm)
{
    int64_t s = start; int lsh = 0;
    if (s < 0) { lsh = (int)(-s); s = 0; }
    if (s > tot) s = tot;
    const int64_t wi = s >> 4; const int sh = (int)(s & 15) * 2;
    const uint64_t lo = (uint64_t)seq2[wi] | (uint64_t)seq2[wi + 1] << 32, hi = seq2[wi + 2];
    uint64_t v = lo >> sh;
    if (sh) v |= hi << (64 - sh);
    const int64_t ni = s >> 5;
    uint32_t nn = (uint32_t)(((uint64_t)nmask[ni] | (uint64_t)nmask[ni + 1] << 32) >> (int)(s & 31));
    if (lsh) { v = lsh >= 32 ? 0 : v << (2 * lsh); nn = lsh >= 32 ? 0 : nn << lsh; }
    w = v; nm = nn;
}
// next 32 bases in consumption order, starting at DP index x of a sequence whose DP index 0 is absolute base i0
__device__ __forceinline__ void d_stream_fill(BaseStream &S, const uint32_t *__restrict__ seq2, const uint32_t *__restrict__ nmask,
                                              int64_t i0, int step, int x, int comp, int64_t tot)
{
    if (step > 0) d_load32(seq2, nmask, i0 + x, tot, S.w, S.nm);
    else { uint64_t w; uint32_t nm; d_load32(seq2, nmask, i0 - x - 31, tot, w, nm); S.w = d_rev2(w, 32); S.nm = __brev(nm); }
    if (comp) S.w = ~S.w;
}
// the same without the ambiguity mask (windows known to be free of N: only cells outside the matrix can see one)
__device__ __forceinline__ void d_stream_fill_acgt(BaseStream &S, const uint32_t *__restrict__ seq2, int64_t i0, int step, int x, int comp, int64_t tot)
{
    int64_t s = step > 0 ? i0 + x : i0 - x - 31; int lsh = 0;
    if (s < 0) { lsh = (int)(-s); s = 0; }
    if (s > tot) s = tot;
    const int64_t wi = s >> 4; const int sh = (int)(s & 15) * 2;
    const uint64_t lo = (uint64_t)seq2[wi] | (uint64_t)seq2[wi + 1] << 32, hi = seq2[wi + 2];
    uint64_t v = lo >> sh;
    if (sh) v |= hi << (64 - sh);
    if (lsh) v = lsh >= 32 ? 0 : v << (2 * lsh);
    if (step <= 0) v = d_rev2(v, 32);
    S.w = comp ? ~v : v; S.nm = 0;
}
__device__ __forceinline__ int d_stream_next_acgt(BaseStream &S) { const int c = (int)((uint32_t)S.w & 3u); S.w >>= 2; return c; }
__device__ __forceinline__ int d_stream_next(BaseStream &S)
{
    const int c = (int)((uint32_t)S.w & 3u), isn = (int)(S.nm & 1u);
    S.w >>= 2; S.nm >>= 1;
    return isn ? 4 : c;
}
__device__ __forceinline__ void d_init_diag(const DpOpt &o, int d, int dhi, int m, int n, bool have,
                                            int32_t &H, int32_t &E1, int32_t &E2, int32_t &F1, int32_t &F2, int32_t &alo, int32_t &span)
{
    H = E1 = E2 = F1 = F2 = TELR_NEG; alo = 0x40000000; span = 0;
    if (!have || d > dhi) return;
    if (d == 0) H = 0;
    else if (o.cx_scale) {
        if (d > 0) { E1 = -d_cx_cost(o, d); E2 = d < o.cx_flat ? d : o.cx_flat; H = E1; }
        else { F1 = -d_cx_cost(o, -d); F2 = -d < o.cx_flat ? -d : o.cx_flat; H = F1; }
    }
    else if (d > 0) { E1 = -(o.q + d * o.e); E2 = -(o.q2 + d * o.e2); H = E1 > E2 ? E1 : E2; }
    else { F1 = -(o.q - d * o.e); F2 = -(o.q2 - d * o.e2); H = F1 > F2 ? F1 : F2; }
    const int lo = (d < 0 ? -d : d) + 2, hi1 = 2 * m + d, hi2 = 2 * n - d, hi = hi1 < hi2 ? hi1 : hi2;
    if (hi >= lo) { alo = lo; span = hi - lo; }
}
// cell without the NEG clamps
__device__ __forceinline__ uint32_t d_cell_nc(const DpOpt &o, int32_t hd, int32_t hl, int32_t e1l, int32_t e2l, int32_t hu, int32_t f1u, int32_t f2u,
                                              int qb, int tbv, int32_t &h, int32_t &ve1, int32_t &vf1, int32_t &ve2, int32_t &vf2)
{
    uint32_t t = 0; int32_t op, g;
    if (o.cx_scale) return d_cell(o, hd, hl, e1l, e2l, hu, f1u, f2u, qb, tbv, h, ve1, vf1, ve2, vf2);      // convex cost: the one scalar cell
    op = hl - (o.q + o.e);   g = e1l - o.e;  ve1 = g > op ? g : op; t |= g > op ? 8u : 0u;
    op = hu - (o.q + o.e);   g = f1u - o.e;  vf1 = g > op ? g : op; t |= g > op ? 16u : 0u;
    op = hl - (o.q2 + o.e2); g = e2l - o.e2; ve2 = g > op ? g : op; t |= g > op ? 32u : 0u;
    op = hu - (o.q2 + o.e2); g = f2u - o.e2; vf2 = g > op ? g : op; t |= g > op ? 64u : 0u;
    int sc = qb == tbv ? o.a : -o.b;
    t |= qb == tbv ? 128u : 0u;
    if ((qb | tbv) > 3) { sc = -o.sc_ambi; t &= ~128u; }
    h = hd + sc; uint32_t src = 0;
    if (ve1 > h) { h = ve1; src = 1; }
    if (vf1 > h) { h = vf1; src = 2; }
    if (ve2 > h) { h = ve2; src = 3; }
    if (vf2 > h) { h = vf2; src = 4; }
    return t | src;
}
#define DPP_SHR1(old, v) __builtin_amdgcn_update_dpp((old), (v), 0x138, 0xf, 0xf, false)   /* lane i <- lane i-1 */
#define DPP_SHL1(old, v) __builtin_amdgcn_update_dpp((old), (v), 0x130, 0xf, 0xf, false)   /* lane i <- lane i+1 */

// NW > 1: one problem per workgroup of NW waves (LPP = 64*NW); the three values that cross a wave boundary per
// step go through LDS with one barrier per step (same scheme as the packed kernel).
template <int LPP, int R, int NW = 1>
__global__ void __launch_bounds__(64 * NW) k_dp_reg(DpArgs A)
{
    static_assert(NW == 1 || LPP == 64 * NW, "multi-wave problems use whole waves");
    constexpr int PPW = NW > 1 ? 1 : 64 / LPP, SLOTS = LPP * R;
    __shared__ int32_t xch[NW > 1 ? 2 * NW * 3 : 1];
    __shared__ int32_t ncell_sh;
    if (R >= 2 || NW > 1) __builtin_amdgcn_s_setprio(3);   // tail classes: few long waves, give them issue priority over the bulk
    const int lane = threadIdx.x, sub = lane / LPP, l = lane % LPP;
    const int wv = NW > 1 ? lane >> 6 : 0, wl = lane & 63;
    const int pi = blockIdx.x * PPW + sub;
    const bool have = pi < A.nlist;
    const int prob = A.list[have ? pi : 0];
    const DpProb P = A.probs[prob];
    const DpOpt o = A.o;
    const int m = have ? P.m : 0, n = have ? P.n : 0, dlo = P.dlo, dhi = P.dhi;
    const int de0 = dlo + 2 * R * l;
    int32_t He[R], E1e[R], E2e[R], F1e[R], F2e[R], Ho[R], E1o[R], E2o[R], F1o[R], F2o[R];
    int32_t aloE[R], spanE[R], aloO[R], spanO[R];
    int qbs[R], tbs[R];
    const int qs_ = P.qstep, ts_ = P.tstep;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        d_init_diag(o, de0 + 2 * r, dhi, m, n, have, He[r], E1e[r], E2e[r], F1e[r], F2e[r], aloE[r], spanE[r]);
        d_init_diag(o, de0 + 2 * r + 1, dhi, m, n, have, Ho[r], E1o[r], E2o[r], F1o[r], F2o[r], aloO[r], spanO[r]);
        // bases of "even step 0" (query rows) and "odd step -1" (target columns)
        const int i0 = ((0 - de0) >> 1) - r, j0 = (de0 >> 1) + r;
        qbs[r] = 4; tbs[r] = 4;
        if (i0 >= 1 && i0 <= m) { int c = d_base(A.qseq2, A.qnmask, P.qi0 + (int64_t)qs_ * (i0 - 1)); qbs[r] = (P.qcomp && c < 4) ? 3 - c : c; }
        if (j0 >= 1 && j0 <= n) tbs[r] = d_base(A.tseq2, A.tnmask, P.ti0 + (int64_t)ts_ * (j0 - 1));
    }
    int amax = m + n;
    if (NW == 1) {
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) { int v = __shfl_xor(amax, s); amax = v > amax ? v : amax; }
    } else {
        if (lane == 0) ncell_sh = 0;
        if (wl == 0) { int32_t *x = xch + (0 * NW + wv) * 3; x[0] = He[0]; x[1] = F1e[0]; x[2] = F2e[0]; }
        __syncthreads();
    }
    uint32_t *tb32 = (uint32_t*)(A.tb + P.tb_off);
    const int last_row = have ? (m + n) >> 2 : -1;
    BaseStream QS, TS; QS.w = 0; QS.nm = 0; TS.w = 0; TS.nm = 0;
    uint32_t tbw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) tbw[r] = 0;
    int ncell = 0;
    for (int a = 1; a <= amax; ++a) {
        const int sh = 8 * (a & 3);
        if (a & 1) {
            // ---- odd step: odd diagonals; one new target base enters at pair R-1
            if ((((a - 1) >> 1) & 31) == 0) d_stream_fill(TS, A.tseq2, A.tnmask, P.ti0, ts_, ((a + de0 + 1) >> 1) + R - 2, 0, A.ttot);
#pragma unroll
            for (int r = 0; r < R - 1; ++r) tbs[r] = tbs[r + 1];
            tbs[R - 1] = d_stream_next(TS);
            int32_t hu = DPP_SHL1(TELR_NEG, He[0]), f1u = DPP_SHL1(TELR_NEG, F1e[0]), f2u = DPP_SHL1(TELR_NEG, F2e[0]);
            if (LPP < 64 && l == LPP - 1) { hu = TELR_NEG; f1u = TELR_NEG; f2u = TELR_NEG; }
            if (NW > 1 && wl == 63 && wv < NW - 1) { const int32_t *x = xch + (0 * NW + wv + 1) * 3; hu = x[0]; f1u = x[1]; f2u = x[2]; }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int32_t uh = r < R - 1 ? He[r < R - 1 ? r + 1 : r] : hu, uf1 = r < R - 1 ? F1e[r < R - 1 ? r + 1 : r] : f1u, uf2 = r < R - 1 ? F2e[r < R - 1 ? r + 1 : r] : f2u;
                if ((uint32_t)(a - aloO[r]) <= (uint32_t)spanO[r]) {
                    int32_t h, ve1, vf1, ve2, vf2;
                    uint32_t t = d_cell_nc(o, Ho[r], He[r], E1e[r], E2e[r], uh, uf1, uf2, qbs[r], tbs[r], h, ve1, vf1, ve2, vf2);
                    tbw[r] |= t << sh; ++ncell;
                    Ho[r] = h; E1o[r] = ve1; F1o[r] = vf1; E2o[r] = ve2; F2o[r] = vf2;
                }
            }
        } else {
            // ---- even step: even diagonals; one new query base enters at pair 0
            if ((((a >> 1) - 1) & 31) == 0) d_stream_fill(QS, A.qseq2, A.qnmask, P.qi0, qs_, ((a - de0) >> 1) - 1, P.qcomp, A.qtot);
#pragma unroll
            for (int r = R - 1; r > 0; --r) qbs[r] = qbs[r - 1];
            qbs[0] = d_stream_next(QS);
            int32_t hl = DPP_SHR1(TELR_NEG, Ho[R - 1]), e1l = DPP_SHR1(TELR_NEG, E1o[R - 1]), e2l = DPP_SHR1(TELR_NEG, E2o[R - 1]);
            if (LPP < 64 && l == 0) { hl = TELR_NEG; e1l = TELR_NEG; e2l = TELR_NEG; }
            if (NW > 1 && wl == 0 && wv > 0) { const int32_t *x = xch + (1 * NW + wv - 1) * 3; hl = x[0]; e1l = x[1]; e2l = x[2]; }
#pragma unroll
            for (int r = R - 1; r >= 0; --r) {
                const int32_t lh = r > 0 ? Ho[r > 0 ? r - 1 : 0] : hl, le1 = r > 0 ? E1o[r > 0 ? r - 1 : 0] : e1l, le2 = r > 0 ? E2o[r > 0 ? r - 1 : 0] : e2l;
                if ((uint32_t)(a - aloE[r]) <= (uint32_t)spanE[r]) {
                    int32_t h, ve1, vf1, ve2, vf2;
                    uint32_t t = d_cell_nc(o, He[r], lh, le1, le2, Ho[r], F1o[r], F2o[r], qbs[r], tbs[r], h, ve1, vf1, ve2, vf2);
                    tbw[r] |= t << sh; ++ncell;
                    He[r] = h; E1e[r] = ve1; F1e[r] = vf1; E2e[r] = ve2; F2e[r] = vf2;
                }
            }
        }
        if (NW > 1) {
            if (a & 1) { if (wl == 63) { int32_t *x = xch + (1 * NW + wv) * 3; x[0] = Ho[R - 1]; x[1] = E1o[R - 1]; x[2] = E2o[R - 1]; } }
            else if (wl == 0) { int32_t *x = xch + (0 * NW + wv) * 3; x[0] = He[0]; x[1] = F1e[0]; x[2] = F2e[0]; }
            __syncthreads();
        }
        if ((a & 3) == 3 || a == amax) {
            if ((a >> 2) <= last_row) {
#pragma unroll
                for (int r = 0; r < R; ++r) tb32[(int64_t)(a >> 2) * SLOTS + l * R + r] = tbw[r];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) tbw[r] = 0;
        }
    }
#pragma unroll
    for (int s = (NW > 1 ? 32 : LPP / 2); s >= 1; s >>= 1) ncell += __shfl_xor(ncell, s);
    if (NW > 1) {
        if (wl == 0) atomicAdd(&ncell_sh, ncell);
        __syncthreads();
        ncell = ncell_sh;
    }
    if (have) {
        const int xf = (n - m) - de0;      // offset of the final diagonal inside this lane's block
        if (xf >= 0 && xf < 2 * R) {
            int32_t sc = TELR_NEG;
#pragma unroll
            for (int r = 0; r < R; ++r) { if (xf == 2 * r) sc = He[r]; if (xf == 2 * r + 1) sc = Ho[r]; }
            DpRes Rr; Rr.score = sc; Rr.bi = m; Rr.bj = n; Rr.nops = 0; Rr.mlen = 0; Rr.cells = ncell; Rr.tbases = n; Rr.mcols = 0;
            A.res[prob] = Rr;
        }
    }
}

// ---- packed-int16 register kernel for short gap fills (classes 10/11).
// Same paired-diagonal scheme as k_dp_reg, but every 32-bit register holds TWO diagonals as
// int16 halves (lane l owns diagonals dlo+4l .. dlo+4l+3: pair 0 in the low halves, pair 1 in the
// high halves), so one v_pk_* instruction advances two cells.  The recurrence runs branch-free over
// the whole band: with H(0,0)=0 and -inf everywhere else, row 0 / column 0 come out of the same
// recurrence (exact for e >= e2, q <= q2, which telr_map enforces) and cells outside the matrix
// only ever see -inf-ish operands, so no activity masks are needed; comparison results are taken
// from sign bits of packed differences.  -inf is -16384 and real scores of these problems stay
// within +-6000 (the class is limited by step count), so int16 never wraps.
// Trace-back bytes: dword tb32[(a>>1)*LPP + l], half (a&1), byte r (pair).
typedef short pk_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk_s2 PKS(uint32_t v) { return __builtin_bit_cast(pk_s2, v); }
__device__ __forceinline__ uint32_t PKU(pk_s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return PKU(PKS(a) + PKS(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return PKU(PKS(a) - PKS(b)); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return PKU(__builtin_elementwise_max(PKS(a), PKS(b))); }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return PKU(__builtin_elementwise_min(PKS(a), PKS(b))); }
__device__ __forceinline__ uint32_t pk_mul(uint32_t a, uint32_t b) { return PKU(PKS(a) * PKS(b)); }
// 0xffff where the half is negative.  Inline asm keeps the packed form: written in C the compiler turns the
// mask-and-select idiom into per-half SDWA compares + v_cndmask + v_perm (about twice the instructions).
__device__ __forceinline__ uint32_t pk_sign(uint32_t a)
{
    uint32_t r;
    asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}
// (m & a) | (~m & b)
__device__ __forceinline__ uint32_t pk_sel(uint32_t m, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_dup(int v) { return ((uint32_t)v & 0xffffu) * 0x00010001u; }
#define PK_NEG 0xC000C000u
// The trace-back spill is written once and read once, milliseconds later, by another kernel: with -DTB_NT the stores and the walk's loads
// are NON-TEMPORAL (streaming through the caches instead of displacing the other range's table lines and filter bitmap from L2 / MALL)
#ifdef TB_NT
typedef uint32_t tb_u2 __attribute__((ext_vector_type(2)));
#define TB_ST2(p, a, b) __builtin_nontemporal_store(tb_u2{(a), (b)}, (tb_u2*)(p))
#define TB_LD2(p) ({ const tb_u2 v_ = __builtin_nontemporal_load((const tb_u2*)(p)); make_uint2(v_.x, v_.y); })
#else
#define TB_ST2(p, a, b) (*(uint2*)(p) = make_uint2((a), (b)))
#define TB_LD2(p) (*(const uint2*)(p))
#endif

struct PkConst { uint32_t qe, e, q2e2, e2, ab, b, a, nab, qeF, q2e2F;      // nab = -(a + b); the nibble cell (TB4) holds them times four, with tags
                 uint32_t cx_oe0, cx_e1, cx_emin, cx_dec; };        // convex cost (d_cell_pk_cx): open + ext(0), ext(1), the extension's floor, its decay

// ONEP (one-piece): in a band of D diagonals no gap run is longer than D - 1, and while (D - 1)(e - e2) < q2 - q the second
// affine piece q2 + L e2 is STRICTLY dearer than q + L e for every possible run length, so E2 / F2 are strictly below
// E1 / F1 wherever they are finite: they never win the maximum (ties prefer E1 / F1 anyway) and never lie on the path.
// The two states, a third of the cell's instructions, are then simply not computed -- bit-identical results (the oracle
// always computes all five states).  map-ont / map-pb: D <= 20, i.e. classes 17 and 10 = 72 % of the fill cells.
template <bool ONEP = false>
__device__ __forceinline__ uint32_t d_cell_pk(const PkConst &c, uint32_t hd, uint32_t hl, uint32_t e1l, uint32_t e2l, uint32_t hu, uint32_t f1u, uint32_t f2u,
                                              uint32_t qb, uint32_t tbv, uint32_t &h, uint32_t &ve1, uint32_t &vf1, uint32_t &ve2, uint32_t &vf2)
{
    uint32_t op, g, t;
    op = pk_sub(hl, c.qe);   g = pk_sub(e1l, c.e);  ve1 = pk_max(op, g); t  = pk_sign(pk_sub(op, g)) & 0x00080008u;
    op = pk_sub(hu, c.qe);   g = pk_sub(f1u, c.e);  vf1 = pk_max(op, g); t |= pk_sign(pk_sub(op, g)) & 0x00100010u;
    if (!ONEP) {
        op = pk_sub(hl, c.q2e2); g = pk_sub(e2l, c.e2); ve2 = pk_max(op, g); t |= pk_sign(pk_sub(op, g)) & 0x00200020u;
        op = pk_sub(hu, c.q2e2); g = pk_sub(f2u, c.e2); vf2 = pk_max(op, g); t |= pk_sign(pk_sub(op, g)) & 0x00400040u;
    } else { ve2 = PK_NEG; vf2 = PK_NEG; }
    // no ambiguity case here: problems with an N inside either window never reach the packed classes (k_prob_sizes)
    const uint32_t eq = pk_sign(pk_sub(qb ^ tbv, 0x00010001u));          // bases equal
    const uint32_t sc = pk_sub(eq & c.ab, c.b);
    t |= eq & 0x00800080u;
    h = pk_add(hd, sc);
    uint32_t m, src;
    m = pk_sign(pk_sub(h, ve1)); h = pk_max(h, ve1); src = m & 0x00010001u;
    m = pk_sign(pk_sub(h, vf1)); h = pk_max(h, vf1); src = pk_sel(m, 0x00020002u, src);
    if (!ONEP) {
        m = pk_sign(pk_sub(h, ve2)); h = pk_max(h, ve2); src = pk_sel(m, 0x00030003u, src);
        m = pk_sign(pk_sub(h, vf2)); h = pk_max(h, vf2); src = pk_sel(m, 0x00040004u, src);
    }
    return t | src;
}

// The cell of the CONVEX gap cost (DpOpt.cx_*; ngmlr-*): one E and one F state, and in the registers that hold E2 / F2 elsewhere
// what the NEXT base of the gap the state ends with costs, x = ext(length) (it travels exactly like its state: same neighbour
// shifts, same lane exchange).  The oracle and the scalar cell keep the length itself; ext(len + 1) = max(emin, ext(len) - dec), so
// carrying the cost is the same function without the multiplication and the clamps:
//   E = max(H_left - open - ext(0)  [opened: x = ext(1)],  E_left - x_left  [extended: x = max(emin, x_left - dec)])
// Every value that enters an x register from outside the recurrence (initial state, band edge, first / last lane) is emin, never
// PK_NEG: x stays inside [emin, emax] and `-inf - x` cannot wrap.
// Flags as in d_cell_pk (bit 3 / 4: E / F extended, bit 7: bases equal, bits 0-2: source 0 diagonal, 1 E, 2 F): the walks need nothing new.
__device__ __forceinline__ uint32_t d_cell_pk_cx(const PkConst &c, uint32_t hd, uint32_t hl, uint32_t el, uint32_t xel, uint32_t hu, uint32_t fu, uint32_t xfu,
                                                 uint32_t qb, uint32_t tbv, uint32_t &h, uint32_t &ve, uint32_t &vf, uint32_t &nxe, uint32_t &nxf)
{
    uint32_t op, g, m, t;
    op = pk_sub(hl, c.cx_oe0); g = pk_sub(el, xel); ve = pk_max(op, g); m = pk_sign(pk_sub(op, g)); t  = m & 0x00080008u;
    nxe = pk_sel(m, pk_max(pk_sub(xel, c.cx_dec), c.cx_emin), c.cx_e1);
    op = pk_sub(hu, c.cx_oe0); g = pk_sub(fu, xfu); vf = pk_max(op, g); m = pk_sign(pk_sub(op, g)); t |= m & 0x00100010u;
    nxf = pk_sel(m, pk_max(pk_sub(xfu, c.cx_dec), c.cx_emin), c.cx_e1);
    const uint32_t eq = pk_sign(pk_sub(qb ^ tbv, 0x00010001u));          // bases equal
    const uint32_t sc = pk_sub(eq & c.ab, c.b);
    t |= eq & 0x00800080u;
    h = pk_add(hd, sc);
    uint32_t src;
    m = pk_sign(pk_sub(h, ve)); h = pk_max(h, ve); src = m & 0x00010001u;
    m = pk_sign(pk_sub(h, vf)); h = pk_max(h, vf); src = pk_sel(m, 0x00020002u, src);
    return t | src;
}

// The one-piece cell of the nibble classes (d_tb4), with PROVENANCE TAGS instead of sign extractions.  Scores are kept times
// four, so the two low bits of a value are free: the candidates of a maximum carry a tag there and the maximum itself tells
// where it came from -- a flag costs an AND of the result instead of a subtraction, a sign extraction and a mask.
//   E = max(H_left - 4(q+e) + 2   [tag 2: opened],  E_left - 4e   [tag 1: E states are stored with tag 1])
//   F = max(H_up   - 4(q+e) + 1   [tag 1: opened],  F_up   - 4e   [tag 0])
//   H = max(H_diag + 4 sc + 2     [tag 2],  E [tag 1],  F [tag 0])        ties: the larger tag wins = diagonal > E > F, and
//                                                                          "opened" beats "extended", as the oracle decides
// The nibble of the cell (bits SH..SH+3; SH = 0 on even steps, 4 on odd steps, so that OR-ing the two steps of a row gives
// one byte per diagonal) is RAW: tag of H (2 = diagonal, 1 = E, 0 = F), bit 0 of E's tag (1 = extended), bit 0 of F's tag
// (0 = extended); the walk turns it into source / extension flags.  No "bases equal" bit: the walk derives the matching
// columns from the score.  A quarter of the int16 range remains: k_prob_sizes sends longer problems to the two-lane class.
template <int SH>
__device__ __forceinline__ uint32_t d_cell_pk4(const PkConst &c, uint32_t hd, uint32_t hl, uint32_t e1l, uint32_t hu, uint32_t f1u,
                                               uint32_t qb, uint32_t tbv, uint32_t &h, uint32_t &ve1, uint32_t &vf1)
{
    const uint32_t et = pk_max(pk_sub(hl, c.qe), pk_sub(e1l, c.e));            // c.qe = 4(q+e) - 2, c.e = 4e
    const uint32_t ft = pk_max(pk_sub(hu, c.qeF), pk_sub(f1u, c.e));           // c.qeF = 4(q+e) - 1
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ve1) : "v"(et), "v"(0xFFFCFFFCu), "v"(0x00010001u));
    vf1 = ft & 0xFFFCFFFCu;
    // diagonal move: min(q ^ t, 1) is the mismatch bit, h = (hd + 4a + 2) - 4(a + b) * mismatch one packed multiply-add
    uint32_t ne, hda = pk_add(hd, c.a), ht;                                    // c.a = 4a + 2, c.nab = -4(a + b)
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(ne) : "v"(qb ^ tbv), "v"(0x00010001u));
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(ht) : "v"(ne), "v"(c.nab), "v"(hda));
    ht = pk_max(pk_max(ht, ve1), vf1);
    h = ht & 0xFFFCFFFCu;
    uint32_t n = ht & 0x00030003u;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(n) : "v"(et & 0x00010001u), "v"(2), "v"(n));
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(n) : "v"(ft & 0x00010001u), "v"(3), "v"(n));
    return SH ? n << SH : n;
}

// The two-piece cell with provenance tags (TAG8): scores times EIGHT, three tag bits.  States are stored under the tag they
// carry as candidates of H: E1 3, F1 2, E2 1, F2 0; the diagonal move has tag 4; an "opened" gap candidate has the stored tag
// of its state plus one.  Ties therefore resolve as in the oracle (diagonal > E1 > F1 > E2 > F2, opened before extended).
// The byte of the cell is RAW: tag of H in bits 0-2 (source = 4 - tag), then bit 0 of the tags of E1, F1, E2, F2 (E: 1 =
// extended, F: 0 = extended); same positions as the flags of d_cell_pk, the walk inverts what needs inverting.  No "bases
// equal" bit: the walk derives the matching columns from the score and the gap runs.  An eighth of the int16 range remains:
// a wave takes this cell only when its longest problem fits (k_dp_pk), else d_cell_pk.
__device__ __forceinline__ uint32_t d_cell_pk8(const PkConst &c, uint32_t hd, uint32_t hl, uint32_t e1l, uint32_t e2l, uint32_t hu, uint32_t f1u, uint32_t f2u,
                                               uint32_t qb, uint32_t tbv, uint32_t &h, uint32_t &ve1, uint32_t &vf1, uint32_t &ve2, uint32_t &vf2)
{
    const uint32_t et1 = pk_max(pk_sub(hl, c.qe), pk_sub(e1l, c.e));           // c.qe = 8(q+e) - 4      opened: tag 4, extended: 3
    const uint32_t ft1 = pk_max(pk_sub(hu, c.qeF), pk_sub(f1u, c.e));          // c.qeF = 8(q+e) - 3     opened: 3, extended: 2
    const uint32_t et2 = pk_max(pk_sub(hl, c.q2e2), pk_sub(e2l, c.e2));        // c.q2e2 = 8(q2+e2) - 2  opened: 2, extended: 1
    const uint32_t ft2 = pk_max(pk_sub(hu, c.q2e2F), pk_sub(f2u, c.e2));       // c.q2e2F = 8(q2+e2) - 1 opened: 1, extended: 0
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ve1) : "v"(et1), "v"(0xFFF8FFF8u), "v"(0x00030003u));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(vf1) : "v"(ft1), "v"(0xFFF8FFF8u), "v"(0x00020002u));
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ve2) : "v"(et2), "v"(0xFFF8FFF8u), "v"(0x00010001u));
    vf2 = ft2 & 0xFFF8FFF8u;
    uint32_t ne, hda = pk_add(hd, c.a), ht;                                    // c.a = 8a + 4, c.nab = -8(a + b)
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(ne) : "v"(qb ^ tbv), "v"(0x00010001u));
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(ht) : "v"(ne), "v"(c.nab), "v"(hda));
    ht = pk_max(pk_max(ht, ve1), pk_max(vf1, pk_max(ve2, vf2)));
    h = ht & 0xFFF8FFF8u;
    uint32_t n = ht & 0x00070007u;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(n) : "v"(et1 & 0x00010001u), "v"(3), "v"(n));
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(n) : "v"(ft1 & 0x00010001u), "v"(4), "v"(n));
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(n) : "v"(et2 & 0x00010001u), "v"(5), "v"(n));
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(n) : "v"(ft2 & 0x00010001u), "v"(6), "v"(n));
    return n;
}

// Lane l of a problem owns the 4R consecutive diagonals dlo+4R*l .. dlo+4R*l+4R-1 as R packed register pairs:
// register r holds {low half: diagonal de0+4r, high half: de0+4r+2} in the "even" set and {+1, +3} in the "odd" set.
// Every register of a lane advances two cells with one packed instruction sequence and the per-step costs (loop
// control, base streams, cross-lane shifts, store) are shared by the R registers, so the kernel is instantiated with
// few lanes and many registers per problem: LPP = 1 (a whole problem per lane, 64 problems per wave, no cross-lane
// traffic at all) for bands up to 32 diagonals, LPP = 2 / 4 above.  The loop is unrolled by two steps (even step 2k,
// odd step 2k+1), which is exactly one row of trace-back dwords: tb32[k*RW + l*R + r], RW = LPP*R, bytes
// {even pair 0, even pair 1, odd pair 0, odd pair 1}.  (That is the LOGICAL byte offset inside a problem's matrix; for LPP = 1 the
// 64 problems of a wave are interleaved in 8-byte units in memory: d_tb_interleaved, k_tb_gather.)
template <int R> __device__ __forceinline__ void d_push_q(uint32_t (&qb)[R], uint32_t v)
{
#pragma unroll
    for (int r = R - 1; r >= 1; --r) qb[r] = __builtin_amdgcn_alignbit(qb[r], qb[r - 1], 16);
    qb[0] = (qb[0] << 16) | v;
}
template <int R> __device__ __forceinline__ void d_push_t(uint32_t (&tb)[R], uint32_t v)
{
#pragma unroll
    for (int r = 0; r < R - 1; ++r) tb[r] = __builtin_amdgcn_alignbit(tb[r + 1], tb[r], 16);
    tb[R - 1] = (tb[R - 1] >> 16) | (v << 16);
}

// EXT = z-drop extension from (0,0) instead of a global fill (oracle band_dp, ext=1): after every step the maximum H
// of the anti-diagonal (smallest diagonal among ties) updates the best cell and the z-drop test may retire the problem.
// Bases past the end of a window carry bit 3 in their code, which keeps the cells beyond row m / column n out of the
// maximum (they never feed a cell inside the matrix).
template <int LPP, int R> __device__ __forceinline__ int d_pk_rowmax(const uint32_t (&hv)[R])
{
    uint32_t M = hv[0];
#pragma unroll
    for (int r = 1; r < R; ++r) M = pk_max(M, hv[r]);
    int lo = (int)(short)(M & 0xffffu), hi = (int)M >> 16, cur = lo > hi ? lo : hi;
#pragma unroll
    for (int s = 1; s < LPP; s <<= 1) { const int v = __shfl_xor(cur, s); cur = v > cur ? v : cur; }
    return cur;
}
// smallest diagonal of the problem whose (valid) H equals cur; par = parity of the step
template <int LPP, int R> __device__ __forceinline__ int d_pk_argd(const uint32_t (&hv)[R], int cur, int de0, int par)
{
    int d = 0x7fffffff;
#pragma unroll
    for (int r = R - 1; r >= 0; --r) {
        const int lo = (int)(short)(hv[r] & 0xffffu), hi = (int)hv[r] >> 16;
        if (hi == cur) d = de0 + 4 * r + 2 + par;
        if (lo == cur) d = de0 + 4 * r + par;
    }
#pragma unroll
    for (int s = 1; s < LPP; s <<= 1) { const int v = __shfl_xor(d, s); d = v < d ? v : d; }
    return d;
}
// interior cells (i >= 1, j >= 1) of anti-diagonal a inside the band and the matrix
__device__ __forceinline__ int d_step_cells(int a, int m, int n, int dlo, int dhi)
{
    int lo = dlo > a - 2 * m ? dlo : a - 2 * m; if (2 - a > lo) lo = 2 - a;
    int hi = dhi < 2 * n - a ? dhi : 2 * n - a; if (a - 2 < hi) hi = a - 2;
    if ((lo - a) & 1) ++lo;
    return hi >= lo ? ((hi - lo) >> 1) + 1 : 0;
}

// NW > 1: one problem per workgroup of NW waves (LPP = 64*NW lanes): the values that cross a wave boundary go
// through `xch` (LDS, [2][NW][3]) with one barrier per step, everything else is unchanged.
// FULL: the first FULL registers of a lane are inside the band for every problem of the class (classes are cut so that
// only the last register can straddle dhi), so they need no out-of-band masks.
template <int LPP, int R, bool EXT, int NW = 1, int FULL = 0, bool ONEP = false, bool TB4 = false, bool TAG8 = false, bool CX = false>
__device__ __forceinline__ void d_dp_pkr(const DpArgs &A, const int32_t *__restrict__ list, int nlist, int first_prob, uint32_t *xch = nullptr)
{
    static_assert(NW == 1 || LPP == 64 * NW, "multi-wave problems use whole waves");
    static_assert(!TB4 || (ONEP && LPP == 1 && !EXT), "nibble spill: one-piece cell, one problem per lane");
    static_assert(!TAG8 || (!ONEP && !TB4 && !EXT && NW == 1), "tagged two-piece cell: fills of the one-launch classes");
    static_assert(!CX || (!ONEP && !TB4 && !TAG8), "convex cell: plain flags, byte spill");
    constexpr int RW = LPP * R;
    const int lane = threadIdx.x, sub = lane / LPP, l = lane % LPP;
    const int wv = NW > 1 ? lane >> 6 : 0, wl = lane & 63;
    const int pi = first_prob + sub;
    const bool have = pi < nlist;
    const int prob = list[have ? pi : 0];
    const DpProb P = A.probs[prob];
    const DpOpt o = A.o;
    PkConst c; c.qe = pk_dup(o.q + o.e); c.e = pk_dup(o.e); c.q2e2 = pk_dup(o.q2 + o.e2); c.e2 = pk_dup(o.e2);
    c.ab = pk_dup(o.a + o.b); c.b = pk_dup(o.b); c.a = pk_dup(o.a); c.nab = pk_dup(-(o.a + o.b)); c.qeF = c.qe;
    c.q2e2F = c.q2e2;
    c.cx_oe0 = pk_dup(o.cx_open + d_cx_ext(o, 0)); c.cx_e1 = pk_dup(d_cx_ext(o, 1)); c.cx_emin = pk_dup(o.cx_emin); c.cx_dec = pk_dup(o.cx_dec);
    const uint32_t XN = CX ? c.cx_emin : PK_NEG;        // what an E2 / F2 register holds outside the recurrence (convex cost: see d_cell_pk_cx)
    if constexpr (TB4) {           // scores times four, provenance tags in the two low bits (d_cell_pk4)
        c.qe = pk_dup(4 * (o.q + o.e) - 2); c.qeF = pk_dup(4 * (o.q + o.e) - 1); c.e = pk_dup(4 * o.e);
        c.a = pk_dup(4 * o.a + 2); c.nab = pk_dup(-4 * (o.a + o.b));
    }
    if constexpr (TAG8) {          // scores times eight, three tag bits (d_cell_pk8)
        c.qe = pk_dup(8 * (o.q + o.e) - 4); c.qeF = pk_dup(8 * (o.q + o.e) - 3); c.e = pk_dup(8 * o.e);
        c.q2e2 = pk_dup(8 * (o.q2 + o.e2) - 2); c.q2e2F = pk_dup(8 * (o.q2 + o.e2) - 1); c.e2 = pk_dup(8 * o.e2);
        c.a = pk_dup(8 * o.a + 4); c.nab = pk_dup(-8 * (o.a + o.b));
    }
    const int m = have ? P.m : 0, n = have ? P.n : 0, dlo = P.dlo;
    const int de0 = dlo + 4 * R * l;                   // lowest (even) diagonal of this lane
    const int dhi = have ? P.dhi : dlo - 1;            // diagonals above dhi are outside the band: their H / F stay -inf
    uint32_t He[R], E1e[R], E2e[R], F1e[R], F2e[R], Ho[R], E1o[R], E2o[R], F1o[R], F2o[R], inE[R], inO[R], qb[R], tbv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        He[r] = E1e[r] = F1e[r] = Ho[r] = E1o[r] = F1o[r] = PK_NEG; E2e[r] = F2e[r] = E2o[r] = F2o[r] = XN;
        if constexpr (TB4) { E1e[r] |= 0x00010001u; E1o[r] |= 0x00010001u; }        // E states carry tag 1
        if constexpr (TAG8) { E1e[r] |= 0x00030003u; E1o[r] |= 0x00030003u; F1e[r] |= 0x00020002u; F1o[r] |= 0x00020002u; E2e[r] |= 0x00010001u; E2o[r] |= 0x00010001u; }
        const int d0 = de0 + 4 * r;
        if (d0 == 0) He[r] &= 0xffff0000u;             // H(0,0) = 0
        if (d0 + 2 == 0) He[r] &= 0x0000ffffu;
        inE[r] = (d0 <= dhi ? 0x0000ffffu : 0u) | (d0 + 2 <= dhi ? 0xffff0000u : 0u);
        inO[r] = (d0 + 1 <= dhi ? 0x0000ffffu : 0u) | (d0 + 3 <= dhi ? 0xffff0000u : 0u);
        qb[r] = 0; tbv[r] = 0;
    }
    const int mn = m + n;
    int amax = mn, amin = have ? mn : 0x7fffffff;
    if (NW == 1) {
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) { int v = __shfl_xor(amax, s); amax = v > amax ? v : amax; v = __shfl_xor(amin, s); amin = v < amin ? v : amin; }
    }
    amin = __builtin_amdgcn_readfirstlane(amin);
    uint32_t *tb32 = (uint32_t*)(A.tb + P.tb_off) + (EXT ? ((l * R) >> 2) * 16 + ((l * R) & 3) : l * R);      // (EXT: the tiled layout below)
    const int last_row = have ? mn >> 1 : -1;
    // base streams: the newest query base enters register 0 (low half) and ages towards register R-1, the newest
    // target base enters register R-1 (high half) and ages towards register 0
    const int qs_ = P.qstep, ts_ = P.tstep;
    BaseStream QS, TS;
    d_stream_fill_acgt(QS, A.qseq2, P.qi0, qs_, -(de0 >> 1) - 2 * R, P.qcomp, A.qtot);
    d_stream_fill_acgt(TS, A.tseq2, P.ti0, ts_, (de0 >> 1) - 1, 0, A.ttot);
#pragma unroll
    for (int z = 0; z < 2 * R; ++z) {
        uint32_t qv = (uint32_t)d_stream_next_acgt(QS), tv = (uint32_t)d_stream_next_acgt(TS);
        if (EXT) { if (-(de0 >> 1) - 2 * R + z >= m) qv |= 8u; if ((de0 >> 1) - 1 + z >= n) tv |= 8u; }
        d_push_q<R>(qb, qv); d_push_t<R>(tbv, tv);
    }
    int qleft = 32 - 2 * R, tleft = 32 - 2 * R;
    int best = 0, bi = 0, bj = 0, prev_cur = -16384, done = have ? 0 : 1, ncell = 0;
    const int zdrop = o.zdrop;
    // RE-BIASING (convex cost: its scores are in 1/cx_scale units, ten to twenty times the affine presets', and a 2,000-base window
    // would leave int16): every 32 trips the problem's row maximum is brought back to zero by moving all its H / E / F values by it and
    // keeping the sum of the moves (`off`); in between the maximum drifts by at most 32 (a + b) S / 2 < 4,000.  Cells more than 12,000
    // below the row maximum become -inf: a path through such a cell can be replaced by one through the row's best cell that joins it
    // later -- one gap across the band and the matches skipped meanwhile, < 5,500 for 128 diagonals at scale 20 -- so it is never the
    // optimum: scores and paths stay those of the oracle's int32 cells.  (A live cell sinks by at most 2 (open + ext) <= 400 a trip:
    // -12,000 - 32 x 400 stays inside int16 until the next round marks it.)  Multi-wave problems take the maximum over their waves
    // through LDS; which band widths the window holds at a preset's scale is the host's decision (telr_engine.hip: pk_cx_ok).
    constexpr bool REB = CX;
    int off = 0, fin_off = 0;
    const bool first = l == 0, last = l == LPP - 1;
    // which register / half holds the final diagonal n-m (its parity is the parity of m+n)
    const int xf = (n - m) - de0;
    const bool fin_here = have && xf >= 0 && xf < 4 * R;
    const int fin_r = xf >> 2, fin_hi = (xf >> 1) & 1;
    uint32_t fin = PK_NEG, te[R], prow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { te[r] = 0; prow[r] = 0; }
    if (NW > 1) {       // the first odd step reads the next wave's initial even state
        if (wl == 0) { uint32_t *x = xch + (0 * NW + wv) * 3; x[0] = He[0]; x[1] = F1e[0]; x[2] = F2e[0]; }
        __syncthreads();
    }
    // step 1 (odd) fills the odd half of row 0; afterwards every trip does the even step 2k and the odd step 2k+1
    for (int k = 0; 2 * k <= amax; ++k) {
        uint32_t h, ve1, vf1, ve2, vf2;
        if (k > 0) {
            const int a = 2 * k;
            if (qleft == 0) { d_stream_fill_acgt(QS, A.qseq2, P.qi0, qs_, ((a - de0) >> 1) - 1, P.qcomp, A.qtot); qleft = 32; }
            { uint32_t qv = (uint32_t)d_stream_next_acgt(QS); if (EXT && ((a - de0) >> 1) - 1 >= m) qv |= 8u; d_push_q<R>(qb, qv); } --qleft;
            uint32_t ph = PK_NEG, pe1 = PK_NEG, pe2 = XN;
            if (LPP > 1) {
                ph = DPP_SHR1((int)PK_NEG, (int)Ho[R - 1]); pe1 = DPP_SHR1((int)PK_NEG, (int)E1o[R - 1]); pe2 = DPP_SHR1((int)XN, (int)E2o[R - 1]);
                if (NW > 1 && wl == 0 && wv > 0) { const uint32_t *x = xch + (1 * NW + wv - 1) * 3; ph = x[0]; pe1 = x[1]; pe2 = x[2]; }
                if (first) { ph = PK_NEG; pe1 = PK_NEG; pe2 = XN; }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // left neighbour: odd diagonal below -> {low: previous register's high half, high: own low half}
                const uint32_t lh = r ? Ho[r - 1] : ph, le1 = r ? E1o[r - 1] : pe1, le2 = r ? E2o[r - 1] : pe2;
                const uint32_t hl = __builtin_amdgcn_alignbit(Ho[r], lh, 16), e1l = __builtin_amdgcn_alignbit(E1o[r], le1, 16), e2l = __builtin_amdgcn_alignbit(E2o[r], le2, 16);
                if constexpr (TB4) { te[r] = d_cell_pk4<0>(c, He[r], hl, e1l, Ho[r], F1o[r], qb[r], tbv[r], h, ve1, vf1); ve2 = PK_NEG; vf2 = PK_NEG; }
                else if constexpr (TAG8) te[r] = d_cell_pk8(c, He[r], hl, e1l, e2l, Ho[r], F1o[r], F2o[r], qb[r], tbv[r], h, ve1, vf1, ve2, vf2);
                else if constexpr (CX) te[r] = d_cell_pk_cx(c, He[r], hl, e1l, e2l, Ho[r], F1o[r], F2o[r], qb[r], tbv[r], h, ve1, vf1, ve2, vf2);
                else te[r] = d_cell_pk<ONEP>(c, He[r], hl, e1l, e2l, Ho[r], F1o[r], F2o[r], qb[r], tbv[r], h, ve1, vf1, ve2, vf2);
                if (r < FULL) { He[r] = h; F1e[r] = vf1; F2e[r] = vf2; }
                else { He[r] = (h & inE[r]) | (PK_NEG & ~inE[r]); F1e[r] = (vf1 & inE[r]) | (PK_NEG & ~inE[r]); F2e[r] = (vf2 & inE[r]) | (XN & ~inE[r]); }
                E1e[r] = ve1; E2e[r] = ve2;
            }
            if (!EXT && a >= amin) {      // only the last steps of a wave can be some problem's last step (lists are sorted by steps)
#pragma unroll
                for (int r = 0; r < R; ++r) if (a == mn && r == fin_r) { fin = He[r]; fin_off = off; }
            }
            if (NW > 1) {
                if (wl == 0) { uint32_t *x = xch + (0 * NW + wv) * 3; x[0] = He[0]; x[1] = F1e[0]; x[2] = F2e[0]; }
                __syncthreads();
            }
            if (EXT) {
                uint32_t hv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) hv[r] = pk_sel(pk_sign(PKU(PKS(qb[r] | tbv[r]) << (pk_s2)(12))), PK_NEG, He[r]);
                const int cur_rel = d_pk_rowmax<LPP, R>(hv), cur = cur_rel + off;
                if (!done) {
                    if (cur > best) { const int cd = d_pk_argd<LPP, R>(hv, cur_rel, de0, 0); best = cur; bi = (a - cd) >> 1; bj = (a + cd) >> 1; }
                    ncell += d_step_cells(a, m, n, dlo, dhi);
                    if (best - (cur > prev_cur ? cur : prev_cur) > zdrop || a >= mn) done = 1;
                }
                prev_cur = cur;
            }
        }
        {
            const int a = 2 * k + 1;
            if (tleft == 0) { d_stream_fill_acgt(TS, A.tseq2, P.ti0, ts_, ((a + de0 + 1) >> 1) + 2 * R - 2, 0, A.ttot); tleft = 32; }
            { uint32_t tv = (uint32_t)d_stream_next_acgt(TS); if (EXT && ((a + de0 + 1) >> 1) + 2 * R - 2 >= n) tv |= 8u; d_push_t<R>(tbv, tv); } --tleft;
            uint32_t nh = PK_NEG, nf1 = PK_NEG, nf2 = XN;
            if (LPP > 1) {
                nh = DPP_SHL1((int)PK_NEG, (int)He[0]); nf1 = DPP_SHL1((int)PK_NEG, (int)F1e[0]); nf2 = DPP_SHL1((int)XN, (int)F2e[0]);
                if (NW > 1 && wl == 63 && wv < NW - 1) { const uint32_t *x = xch + (0 * NW + wv + 1) * 3; nh = x[0]; nf1 = x[1]; nf2 = x[2]; }
                if (last) { nh = PK_NEG; nf1 = PK_NEG; nf2 = XN; }
            }
            uint32_t row[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // up neighbour: even diagonal above -> {low: own high half, high: next register's low half}
                const uint32_t uh = r < R - 1 ? He[r + 1] : nh, uf1 = r < R - 1 ? F1e[r + 1] : nf1, uf2 = r < R - 1 ? F2e[r + 1] : nf2;
                const uint32_t hu = __builtin_amdgcn_alignbit(uh, He[r], 16), f1u = __builtin_amdgcn_alignbit(uf1, F1e[r], 16), f2u = __builtin_amdgcn_alignbit(uf2, F2e[r], 16);
                uint32_t t;
                if constexpr (TB4) { t = d_cell_pk4<4>(c, Ho[r], He[r], E1e[r], hu, f1u, qb[r], tbv[r], h, ve1, vf1); ve2 = PK_NEG; vf2 = PK_NEG; }
                else if constexpr (TAG8) t = d_cell_pk8(c, Ho[r], He[r], E1e[r], E2e[r], hu, f1u, f2u, qb[r], tbv[r], h, ve1, vf1, ve2, vf2);
                else if constexpr (CX) t = d_cell_pk_cx(c, Ho[r], He[r], E1e[r], E2e[r], hu, f1u, f2u, qb[r], tbv[r], h, ve1, vf1, ve2, vf2);
                else t = d_cell_pk<ONEP>(c, Ho[r], He[r], E1e[r], E2e[r], hu, f1u, f2u, qb[r], tbv[r], h, ve1, vf1, ve2, vf2);
                if (r < FULL) { Ho[r] = h; F1o[r] = vf1; F2o[r] = vf2; }
                else { Ho[r] = (h & inO[r]) | (PK_NEG & ~inO[r]); F1o[r] = (vf1 & inO[r]) | (PK_NEG & ~inO[r]); F2o[r] = (vf2 & inO[r]) | (XN & ~inO[r]); }
                E1o[r] = ve1; E2o[r] = ve2;
                row[r] = TB4 ? (te[r] | t) : __builtin_amdgcn_perm(t, te[r], 0x06040200u);
            }
            if (!EXT && a >= amin) {
#pragma unroll
                for (int r = 0; r < R; ++r) if (a == mn && r == fin_r) { fin = Ho[r]; fin_off = off; }
            }
            if (EXT) {
                uint32_t hv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) hv[r] = pk_sel(pk_sign(PKU(PKS(qb[r] | tbv[r]) << (pk_s2)(12))), PK_NEG, Ho[r]);
                const int cur_rel = d_pk_rowmax<LPP, R>(hv), cur = cur_rel + off;
                if (!done) {
                    if (cur > best) { const int cd = d_pk_argd<LPP, R>(hv, cur_rel, de0, 1); best = cur; bi = (a - cd) >> 1; bj = (a + cd) >> 1; }
                    ncell += d_step_cells(a, m, n, dlo, dhi);
                    if (best - (cur > prev_cur ? cur : prev_cur) > zdrop || a >= mn) done = 1;
                }
                prev_cur = cur;
            }
            if constexpr (TB4) {
                // nibble rows (d_tb4): register pairs -> dwords {r: low-half byte, high-half byte, r+1: ...}; rows k-1 / k leave
                // together as RW4 8-byte units of the wave-interleaved layout (unit u of this lane at tb32 + u*128 dwords)
                constexpr int RW4 = (R + 1) / 2;
                uint32_t rq[RW4];
#pragma unroll
                for (int j = 0; j < RW4; ++j) rq[j] = __builtin_amdgcn_perm(2 * j + 1 < R ? row[2 * j + 1] : 0u, row[2 * j], 0x06040200u);
                uint32_t sq[2 * RW4 + 1];
                if (k & 1) {
#pragma unroll
                    for (int j = 0; j < RW4; ++j) { sq[j] = prow[j]; sq[RW4 + j] = rq[j]; }
                    sq[2 * RW4] = 0;
                    uint32_t *dst = tb32 + (int64_t)((k - 1) / 2 * RW4) * 128;
                    if (k <= last_row) {
#pragma unroll
                        for (int u = 0; u < RW4; ++u) TB_ST2(dst + u * 128, sq[2 * u], sq[2 * u + 1]);
                    } else if (k - 1 <= last_row) {
#pragma unroll
                        for (int u = 0; u < (RW4 + 1) / 2; ++u) TB_ST2(dst + u * 128, sq[2 * u], sq[2 * u + 1]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < RW4; ++j) prow[j] = rq[j];
                    if (2 * (k + 1) > amax && k <= last_row) {        // last trip of the wave: nothing will pair with this row
#pragma unroll
                        for (int j = 0; j < RW4; ++j) sq[j] = rq[j];
                        sq[RW4] = 0;
                        uint32_t *dst = tb32 + (int64_t)(k / 2 * RW4) * 128;
#pragma unroll
                        for (int u = 0; u < (RW4 + 1) / 2; ++u) TB_ST2(dst + u * 128, sq[2 * u], sq[2 * u + 1]);
                    }
                }
            } else if constexpr (LPP == 1 && !EXT) {
                // one lane owns whole rows, and rows k-1 / k are adjacent: they leave as R 8-byte units of the wave-interleaved
                // layout (k_tb_gather): unit u of this lane at tb32 + u*128 dwords, so each store instruction of the wave
                // writes 512 contiguous bytes
                uint32_t sq[2 * R + 1];
                if (k & 1) {
#pragma unroll
                    for (int r = 0; r < R; ++r) { sq[r] = prow[r]; sq[R + r] = row[r]; }
                    sq[2 * R] = 0;
                    uint32_t *dst = tb32 + (int64_t)((k - 1) / 2 * R) * 128;
                    if (k <= last_row) {
#pragma unroll
                        for (int u = 0; u < R; ++u) TB_ST2(dst + u * 128, sq[2 * u], sq[2 * u + 1]);
                    } else if (k - 1 <= last_row) {
#pragma unroll
                        for (int u = 0; u < (R + 1) / 2; ++u) TB_ST2(dst + u * 128, sq[2 * u], sq[2 * u + 1]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) prow[r] = row[r];
                    if (2 * (k + 1) > amax && k <= last_row) {        // last trip of the wave: nothing will pair with this row
#pragma unroll
                        for (int r = 0; r < R; ++r) sq[r] = row[r];
                        sq[R] = 0;
                        uint32_t *dst = tb32 + (int64_t)(k / 2 * R) * 128;
#pragma unroll
                        for (int u = 0; u < (R + 1) / 2; ++u) TB_ST2(dst + u * 128, sq[2 * u], sq[2 * u + 1]);
                    }
                }
            } else if constexpr (EXT) {
                // TILED spill of the extension classes (round 6): four consecutive rows of a lane's 16 bytes (R = 4: the lane's own four
                // registers; R = 1: four neighbouring lanes) share one 64-byte line -- row k of dword x at dword
                // (k >> 2) * 4 RW + (x >> 2) * 16 + (k & 3) * 4 + (x & 3).  The walk (d_traceback_lane) moves up a diagonal, i.e. up the
                // rows in one byte column: one line per FOUR steps instead of one (64-diagonal band) or two (128) per step.
                static_assert(!EXT || R == 4 || R == 1, "tiled spill: a lane owns a whole 16-byte group (R = 4) or one dword of it (R = 1)");
                if (k <= last_row) {
                    uint32_t *dst = tb32 + (int64_t)(k >> 2) * (4 * RW) + (k & 3) * 4;
                    if constexpr (R == 4) *(uint4*)dst = make_uint4(row[0], row[1], row[2], row[3]);
                    else dst[0] = row[0];
                }
            } else if (k <= last_row) {
                uint32_t *dst = tb32 + (int64_t)k * RW;
                if constexpr (R % 4 == 0) {
#pragma unroll
                    for (int r = 0; r < R; r += 4) *(uint4*)(dst + r) = make_uint4(row[r], row[r + 1], row[r + 2], row[r + 3]);
                } else if constexpr (R % 2 == 0) {
#pragma unroll
                    for (int r = 0; r < R; r += 2) *(uint2*)(dst + r) = make_uint2(row[r], row[r + 1]);
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) dst[r] = row[r];
                }
            }
        }
        if constexpr (REB) {
            if ((k & 31) == 31) {
                uint32_t hv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) hv[r] = pk_max(He[r], Ho[r]);
                int mx = d_pk_rowmax<(LPP > 64 ? 64 : LPP), R>(hv);
                if (NW > 1) {             // one problem over NW waves: the maximum of the waves' maxima
                    int32_t *xm = (int32_t*)(xch + 2 * NW * 3);
                    if (wl == 0) xm[wv] = mx;
                    __syncthreads();
#pragma unroll
                    for (int w_ = 0; w_ < NW; ++w_) mx = xm[w_] > mx ? xm[w_] : mx;
                }
                const int delta = mx < -8192 ? 0 : mx;            // (a row of -inf: no problem in these lanes, or one past its end)
                const uint32_t dd = pk_dup(delta), thr = pk_dup(delta - 12000);
#define REB_ONE(x) { const uint32_t dead = pk_sign(pk_sub((x), thr)); (x) = pk_sel(dead, PK_NEG, pk_max(pk_sub((x), dd), PK_NEG)); }
#pragma unroll
                for (int r = 0; r < R; ++r) { REB_ONE(He[r]) REB_ONE(Ho[r]) REB_ONE(E1e[r]) REB_ONE(E1o[r]) REB_ONE(F1e[r]) REB_ONE(F1o[r]) }
#undef REB_ONE
                off += delta;
            }
        }
        if (NW > 1) {
            if (wl == 63) { uint32_t *x = xch + (1 * NW + wv) * 3; x[0] = Ho[R - 1]; x[1] = E1o[R - 1]; x[2] = E2o[R - 1]; }
            __syncthreads();
        }
        if (EXT && __all(done)) break;
    }
    if (EXT) {
        if (have && l == 0) {
            DpRes Rr; Rr.score = best; Rr.bi = bi; Rr.bj = bj; Rr.nops = 0; Rr.mlen = 0; Rr.cells = ncell; Rr.tbases = n; Rr.mcols = 0;
            A.res[prob] = Rr;
        }
        return;
    }
    if (fin_here) {
        int sc = (int)(short)(fin_hi ? (fin >> 16) : (fin & 0xffffu)) + fin_off;
        if constexpr (TB4) sc >>= 2;
        if constexpr (TAG8) sc >>= 3;
        DpRes Rr; Rr.score = sc; Rr.bi = m; Rr.bj = n; Rr.nops = 0; Rr.mlen = 0; Rr.cells = P.pad[1]; Rr.tbases = n; Rr.mcols = 0;
        A.res[prob] = Rr;
    }
}

// All packed classes run as ONE launch: a wave is described by (class, first problem of its class list) and the
// wave table is ordered by decreasing estimated cost (steps x registers per lane), so the long waves start first and
// no class leaves the machine idle behind its own tail.
#define PK_NC 9                            /* packed fill classes of the one cost-ordered launch: 10 .. 17 and 22 */
__host__ __device__ __forceinline__ int PK_CLS(int c) { return c < 8 ? 10 + c : 22; }
__host__ __device__ __forceinline__ int PK_IDX(int cls) { return cls == 22 ? 8 : cls - 10; }
__device__ __constant__ const int PK_LPP[PK_NC] = { 1, 1, 1, 1, 2, 2, 2, 1, 4 };
__device__ __constant__ const int PK_R[PK_NC]   = { 5, 6, 7, 8, 5, 6, 8, 4, 8 };
struct PkPlan { int32_t woff[PK_NC + 1]; };   // first wave of class 10+c in the unsorted wave table
__global__ void k_pk_waves(const DpProb *__restrict__ probs, const int32_t *__restrict__ cls_list, ClsOff off, PkPlan plan,
                           uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plan.woff[PK_NC]) return;
    int c = 0;
    while (i >= plan.woff[c + 1]) ++c;
    const int first = (i - plan.woff[c]) * (64 / PK_LPP[c]);
    const DpProb P = probs[cls_list[off.off[PK_CLS(c)] + first]];     // lists are sorted by decreasing steps
    const uint32_t cost = (uint32_t)((P.m + P.n) * PK_R[c]);          // < 2^16 (packed fills: m + n <= 7,900, at most 8 registers): two radix passes
    keys[i] = cost > 0xFFFFu ? 0xFFFFu : cost;
    vals[i] = (uint32_t)PK_CLS(c) << 26 | (uint32_t)first;
}
#ifndef PK_WPE
#define PK_WPE 2
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PK_WPE))) k_dp_pk(DpArgs A, const uint32_t *__restrict__ waves, const int32_t *__restrict__ cls_list,
                                              ClsOff off)
{
    const uint32_t w = waves[blockIdx.x];
    const int cls = (int)(w >> 26), first = (int)(w & 0x3ffffffu);
    const int32_t *list = cls_list + off.off[cls];
    const int n = off.off[cls + 1] - off.off[cls];
    // widest band in which the second affine piece can never pay (see d_cell_pk): (D - 1)(e - e2) < q2 - q
    if (A.o.cx_scale) {          // convex gap cost: its own cell (plain flags, byte spill) in every class
        switch (cls) {
        case 10: d_dp_pkr<1, 5, false, 1, 4, false, false, false, true>(A, list, n, first); break;
        case 11: d_dp_pkr<1, 6, false, 1, 5, false, false, false, true>(A, list, n, first); break;
        case 12: d_dp_pkr<1, 7, false, 1, 6, false, false, false, true>(A, list, n, first); break;
        case 13: d_dp_pkr<1, 8, false, 1, 7, false, false, false, true>(A, list, n, first); break;
        case 17: d_dp_pkr<1, 4, false, 1, 0, false, false, false, true>(A, list, n, first); break;
        case 14: d_dp_pkr<2, 5, false, 1, 0, false, false, false, true>(A, list, n, first); break;
        case 15: d_dp_pkr<2, 6, false, 1, 0, false, false, false, true>(A, list, n, first); break;
        case 22: d_dp_pkr<4, 8, false, 1, 0, false, false, false, true>(A, list, n, first); break;
        default: d_dp_pkr<2, 8, false, 1, 0, false, false, false, true>(A, list, n, first); break;
        }
        return;
    }
    const int onep_d = d_onep_d(A.o.q, A.o.e, A.o.q2, A.o.e2);
    if (cls == 17 && onep_d >= 16) { if (A.tb4 & 1) d_dp_pkr<1, 4, false, 1, 0, true, true>(A, list, n, first); else d_dp_pkr<1, 4, false, 1, 0, true>(A, list, n, first); return; }
    if (cls == 10 && onep_d >= 20) { if (A.tb4 & 2) d_dp_pkr<1, 5, false, 1, 4, true, true>(A, list, n, first); else d_dp_pkr<1, 5, false, 1, 4, true>(A, list, n, first); return; }
    // two-piece classes: provenance tags (scores times eight) when the wave's longest problem -- its first: the lists are sorted
    // by decreasing steps -- fits an eighth of the int16 range; k_traceback_pk decides the same way
    const DpProb P0 = A.probs[list[first]];
    if (P0.m + P0.n <= A.tag8_steps) {
        switch (cls) {
        case 10: d_dp_pkr<1, 5, false, 1, 4, false, false, true>(A, list, n, first); break;
        case 11: d_dp_pkr<1, 6, false, 1, 5, false, false, true>(A, list, n, first); break;
        case 12: d_dp_pkr<1, 7, false, 1, 6, false, false, true>(A, list, n, first); break;
        case 13: d_dp_pkr<1, 8, false, 1, 7, false, false, true>(A, list, n, first); break;
        case 17: d_dp_pkr<1, 4, false, 1, 0, false, false, true>(A, list, n, first); break;
        case 14: d_dp_pkr<2, 5, false, 1, 0, false, false, true>(A, list, n, first); break;
        case 15: d_dp_pkr<2, 6, false, 1, 0, false, false, true>(A, list, n, first); break;
        case 22: d_dp_pkr<4, 8, false, 1, 0, false, false, true>(A, list, n, first); break;
        default: d_dp_pkr<2, 8, false, 1, 0, false, false, true>(A, list, n, first); break;
        }
        return;
    }
    switch (cls) {
    case 10: d_dp_pkr<1, 5, false, 1, 4>(A, list, n, first); break;      // 17..20 diagonals: registers 0-3 are inside the band
    case 11: d_dp_pkr<1, 6, false, 1, 5>(A, list, n, first); break;
    case 12: d_dp_pkr<1, 7, false, 1, 6>(A, list, n, first); break;
    case 13: d_dp_pkr<1, 8, false, 1, 7>(A, list, n, first); break;
    case 17: d_dp_pkr<1, 4, false>(A, list, n, first); break;            // up to 16 diagonals
    case 14: d_dp_pkr<2, 5, false>(A, list, n, first); break;
    case 15: d_dp_pkr<2, 6, false>(A, list, n, first); break;
    case 22: d_dp_pkr<4, 8, false>(A, list, n, first); break;            // 65..128 diagonals
    default: d_dp_pkr<2, 8, false>(A, list, n, first); break;
    }
}
// wide gap fills in int16 (classes 19-21: bands up to 256 / 512 / 1024 diagonals): one problem per workgroup of
// NW waves, four diagonals per lane; these are few and long, so what counts is the time of one step
template <int NW>
__global__ void __launch_bounds__(64 * NW) k_dp_pkw(DpArgs A)
{
    __shared__ uint32_t xch[2 * NW * 3 + NW];          // values that cross a wave boundary; the waves' row maxima (re-biasing)
    __builtin_amdgcn_s_setprio(3);
    if (A.o.cx_scale) d_dp_pkr<64 * NW, 1, false, NW, 0, false, false, false, true>(A, A.list, A.nlist, blockIdx.x, xch);
    else d_dp_pkr<64 * NW, 1, false, NW>(A, A.list, A.nlist, blockIdx.x, xch);
}
// z-drop extensions (class 18): their own launch on a side stream; four lanes per problem keep the single-wave
// latency of the long windows down
#define PKX_LPP 4
#define PKX_R 4
__global__ void __launch_bounds__(64) k_dp_pkx(DpArgs A)
{
    __builtin_amdgcn_s_setprio(3);
    if (A.o.cx_scale) d_dp_pkr<PKX_LPP, PKX_R, true, 1, 0, false, false, false, true>(A, A.list, A.nlist, blockIdx.x * (64 / PKX_LPP));
    else d_dp_pkr<PKX_LPP, PKX_R, true>(A, A.list, A.nlist, blockIdx.x * (64 / PKX_LPP));
}
// the same class with sixteen lanes per problem (same trace-back layout): a quarter of the work per lane and step, for
// calls with few extensions, where the longest window is what the caller waits for
__global__ void __launch_bounds__(64) k_dp_pkx16(DpArgs A)
{
    __builtin_amdgcn_s_setprio(3);
    if (A.o.cx_scale) d_dp_pkr<16, 1, true, 1, 0, false, false, false, true>(A, A.list, A.nlist, blockIdx.x * 4);
    else d_dp_pkr<16, 1, true>(A, A.list, A.nlist, blockIdx.x * 4);
}

// the wider extension bands (classes 23 / 24: D <= 128 / 256), same cell, same spill layout rule (32 / 64 dwords per row)
#define PKX8_LPP 8           /* lanes per problem of class 23: 8 lanes x 4 registers; 4 x 8 (16 problems per wave, 132 VGPRs) measured slower in round 5: 15.1 vs 11.7 ms per range */
__global__ void __launch_bounds__(64) k_dp_pkx_w8(DpArgs A)
{
    __builtin_amdgcn_s_setprio(3);
    if (A.o.cx_scale) d_dp_pkr<PKX8_LPP, 32 / PKX8_LPP, true, 1, 0, false, false, false, true>(A, A.list, A.nlist, blockIdx.x * (64 / PKX8_LPP));
    else d_dp_pkr<PKX8_LPP, 32 / PKX8_LPP, true>(A, A.list, A.nlist, blockIdx.x * (64 / PKX8_LPP));
}
__global__ void __launch_bounds__(64) k_dp_pkx_w16(DpArgs A)
{
    __builtin_amdgcn_s_setprio(3);
    if (A.o.cx_scale) d_dp_pkr<16, 4, true, 1, 0, false, false, false, true>(A, A.list, A.nlist, blockIdx.x * 4);
    else d_dp_pkr<16, 4, true>(A, A.list, A.nlist, blockIdx.x * 4);
}

// ---- trace-back: one thread per problem walks its trace-back bytes and writes the
// run-length CIGAR in end->start order (64 independent pointer chases per wave).
// `list` (optional) restricts the launch to the problems of one class list.
// The bytes of one problem are visited in (almost) decreasing address order, a few per 64-byte line, so every lane
// keeps the two lines it is walking through in LDS (slot = line parity; dword k of lane l at word k*64+l:
// conflict-free byte reads) and fetches the next lower line into registers ahead of time: memory is touched once
// per line instead of once per step, and a line is normally there before the walk reaches it.
#ifndef TB_SLOTS
#define TB_SLOTS 2     /* 64-byte lines a lane keeps in LDS */
#endif
template <int LAYOUT>       // 2: packed int16 classes (rows of 2 anti-diagonals); -1: take it from the problem's class
__device__ __forceinline__ void d_traceback_lane(const DpProb *__restrict__ probs, DpRes *__restrict__ res, int pi,
                                                 const uint8_t *__restrict__ tb_all, uint32_t *__restrict__ cig, int32_t *__restrict__ retry,
                                                 uint32_t *stage)
{
    const int lane = threadIdx.x;
    const DpProb P = probs[pi];
    if (P.kind >= 3) return;
    const int dhi_ = P.dhi; int touched = 0;
    const int cls = P.pad[0] & 0xff, dlo = P.dlo, mg = P.pad[0] >> 8;
    const int D = P.dhi - dlo + 1, stride = (D + 2) / 2;
    const int lpp = cls >= 5 ? d_cls_slots(cls) : 0;
    const bool packed = cls >= 5, tiled = d_tb_tiled(cls);
    int i = res[pi].bi, j = res[pi].bj;
    uint32_t *cg = cig + P.cig_off;
    int no = 0, ml = 0, mc = 0, state = 0, cur_op = -1, cur_len = 0;
    const int rowb = tiled ? lpp * 16 : packed ? lpp * 4 : stride;      // bytes between consecutive rows of the trace-back matrix (tiled: between tiles of four rows)
    int64_t tag0 = -1, tag1 = -1, pfb = -2;           // lines in the two LDS slots; 128-byte block held in registers
    uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0, v4 = v0, v5 = v0, v6 = v0, v7 = v0;
    while (i > 0 && j > 0) {
        const int a = i + j, sl = (j - i - dlo) >> 1;
        // tiled (the extension classes, d_dp_pkx): row k = a >> 1 of dword x = sl >> 1 at dword (k >> 2) 4 lpp + (x >> 2) 16 + (k & 3) 4 + (x & 3)
        const int64_t off = tiled ? ((((int64_t)(a >> 3) * (4 * lpp) + ((sl >> 3) << 4) + (((a >> 1) & 3) << 2) + ((sl >> 1) & 3)) << 2) + ((a & 1) << 1) + (sl & 1))
                          : (LAYOUT == 2 || cls >= 10) ? ((((int64_t)(a >> 1) * lpp + (sl >> 1)) << 2) + ((a & 1) << 1) + (sl & 1))
                          : packed ? ((((int64_t)(a >> 2) * lpp + sl) << 2) + (a & 3)) : ((int64_t)a * stride + sl);
        // lines are taken relative to the 64-byte grid of the whole scratch buffer; the wave-interleaved classes keep their
        // bytes in 8-byte units 512 bytes apart (k_tb_gather) -- this generic walk reaches them only in the TELR_AB=tb_one_launch mode
        const int64_t abs_off = P.tb_off + (d_tb_interleaved(cls) ? ((off >> 3) << 9) + (off & 7) : off), line = abs_off >> 6;
        const int slot = TB_SLOTS > 1 ? (int)(line & 1) : 0;
        if ((slot ? tag1 : tag0) != line) {
            // Memory is read in whole 128-byte blocks (8 x 16 B per lane): scattered 64-byte reads reach 2.5 TB/s on this
            // part, 128-byte ones 5.4 TB/s (tools/ubench/tb_pattern.hip).  The block stays in registers; its two halves
            // feed the LDS slots one after the other.
            const int64_t blk = line >> 1;
            if (pfb != blk) {
                const uint4 *src = (const uint4*)(tb_all + (blk << 7));
                v0 = src[0]; v1 = src[1]; v2 = src[2]; v3 = src[3]; v4 = src[4]; v5 = src[5]; v6 = src[6]; v7 = src[7];
            }
            uint32_t *dst = stage + slot * 1024 + lane;
            if (line & 1) {
                dst[0 * 64] = v4.x; dst[1 * 64] = v4.y; dst[2 * 64] = v4.z; dst[3 * 64] = v4.w;
                dst[4 * 64] = v5.x; dst[5 * 64] = v5.y; dst[6 * 64] = v5.z; dst[7 * 64] = v5.w;
                dst[8 * 64] = v6.x; dst[9 * 64] = v6.y; dst[10 * 64] = v6.z; dst[11 * 64] = v6.w;
                dst[12 * 64] = v7.x; dst[13 * 64] = v7.y; dst[14 * 64] = v7.z; dst[15 * 64] = v7.w;
            } else {
                dst[0 * 64] = v0.x; dst[1 * 64] = v0.y; dst[2 * 64] = v0.z; dst[3 * 64] = v0.w;
                dst[4 * 64] = v1.x; dst[5 * 64] = v1.y; dst[6 * 64] = v1.z; dst[7 * 64] = v1.w;
                dst[8 * 64] = v2.x; dst[9 * 64] = v2.y; dst[10 * 64] = v2.z; dst[11 * 64] = v2.w;
                dst[12 * 64] = v3.x; dst[13 * 64] = v3.y; dst[14 * 64] = v3.z; dst[15 * 64] = v3.w;
            }
            if (slot) tag1 = line; else tag0 = line;
            // next line to come: the same position one row up when rows are wider than a line, else the line below;
            // fetch its block now unless it is the one already held
            const int64_t nl = rowb >= 64 ? (abs_off - rowb) >> 6 : line - 1, nb = nl >> 1;
            pfb = blk;
            if (nb != blk && nl >= 0) {
                const uint4 *src = (const uint4*)(tb_all + (nb << 7));
                v0 = src[0]; v1 = src[1]; v2 = src[2]; v3 = src[3]; v4 = src[4]; v5 = src[5]; v6 = src[6]; v7 = src[7];
                pfb = nb;
            }
        }
        const int w = (int)(abs_off & 63);
        const uint8_t *st8 = (const uint8_t*)stage + ((slot * 1024 + lane) << 2);       // byte x of this lane's line at st8[(x >> 2) * 256 + (x & 3)]
        const uint32_t t = st8[(w >> 2) * 256 + (w & 3)];
        // tiled: the cells one to three steps up the diagonal are the same byte column of the rows above, 16 bytes apart in this
        // very line while the row is not the tile's first -- read with the current cell, a run of diagonal moves is one trip
        const int mij = i < j ? i : j, kr = tiled ? (a >> 1) & 3 : 0;
        const bool v1 = kr >= 1 && mij > 1, v2 = kr >= 2 && mij > 2, v3 = kr >= 3 && mij > 3;
        const uint32_t t1 = v1 ? st8[((w - 16) >> 2) * 256 + (w & 3)] : 0xffu, t2 = v2 ? st8[((w - 32) >> 2) * 256 + (w & 3)] : 0xffu, t3 = v3 ? st8[((w - 48) >> 2) * 256 + (w & 3)] : 0xffu;
        const uint32_t nd = (uint32_t)((t1 & 7u) != 0u) << 1 | (uint32_t)((t2 & 7u) != 0u) << 2 | (uint32_t)((t3 & 7u) != 0u) << 3;
        const uint32_t mb = (t >> 7) | ((t1 >> 7) & 1u) << 1 | ((t2 >> 7) & 1u) << 2 | ((t3 >> 7) & 1u) << 3;
        touched |= (j - i - dlo <= mg) | (dhi_ - (j - i) <= mg);       // within mg diagonals of a band edge
        // one step of the walk without branches: the 64 lanes are in 64 different states
        const int s0 = state ? state : (int)(t & 7);                 // 0 = diagonal, 1/3 = deletion (E1/E2), 2/4 = insertion (F1/F2)
        const int isM = s0 == 0, isD = s0 & 1;
        const int op = isM ? 0 : (isD ? 2 : 1);
        const int nn = isM ? __builtin_ctz(nd | 16u) : 1;            // cells of this trip: a run of 1 .. 4 diagonal moves, or one step of a gap (nd bit 0 is clear)
        state = (isM || !((t >> (2 + s0)) & 1)) ? 0 : s0;            // a gap state continues while its extension flag is set
        ml += isM ? __builtin_popcount(mb & ((1u << nn) - 1u)) : 0; mc += isM ? nn : 0;
        i -= isM ? nn : (isD ^ 1); j -= isM ? nn : isD;
        const bool same = op == cur_op;
        if (!same && cur_len) cg[no++] = (uint32_t)cur_len << 4 | (uint32_t)cur_op;
        cur_len = same ? cur_len + nn : nn; cur_op = op;
    }
    if (i > 0) { if (cur_op == 1) cur_len += i; else { if (cur_len) cg[no++] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; cur_op = 1; cur_len = i; } }
    if (j > 0) { if (cur_op == 2) cur_len += j; else { if (cur_len) cg[no++] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; cur_op = 2; cur_len = j; } }
    if (cur_len) cg[no++] = (uint32_t)cur_len << 4 | (uint32_t)cur_op;
    res[pi].nops = no; res[pi].mlen = ml; res[pi].mcols = mc;
    if (retry && P.kind == 0 && touched && P.m + P.n <= ADAPT_MAX_STEPS) retry[pi] = 1;
}
__global__ void __launch_bounds__(64) k_traceback(const DpProb *__restrict__ probs, DpRes *__restrict__ res, int32_t np,
                                                  const uint8_t *__restrict__ tb_all, uint32_t *__restrict__ cig, int32_t *__restrict__ retry,
                                                  const int32_t *__restrict__ list)
{
    __shared__ uint32_t stage[TB_SLOTS * 16 * 64];
    const int ti = blockIdx.x * blockDim.x + threadIdx.x;
    if (ti >= np) return;
    d_traceback_lane<-1>(probs, res, list ? list[ti] : ti, tb_all, cig, retry, stage);
}
// Row-synchronous walk for the packed fill classes: one lane per problem, all 64 problems of a wave in the same class
// (same row width), every problem's matrix starting on a 64-byte line.  The plain lane-per-problem walk above stalls on
// memory in almost every step: 64 lanes cross line boundaries at 64 different moments and each crossing waits for the
// wave's youngest load.  Here the WAVE moves down its trace-back matrices one 64-byte line at a time, so all lanes
// cross into a new line in the same iteration, the control flow around the loads is uniform, and
// two register sets hold the next two lines in flight: one memory wait per line for the whole wave instead of one per step.
// rb4 != 0: the class spills nibbles (d_tb4), rb4 = bytes per row; the matching columns then come from the score (o = gap costs)
// tag8: the wave's bytes are the raw tags of d_cell_pk8 (no "bases equal" bit either)
// ENC (round 6): the encoding of the wave's bytes as a COMPILE-TIME choice -- 0 plain flags, 1 raw tags of d_cell_pk8, 2 nibbles: the walk is a chain of
// dependent steps whose issue slots set the wave's time (profiles/r06_traceback_experiments.txt); with the three encodings behind run-time branches its loop
// was 150 instructions and 12 branches.  IL: the wave-interleaved layout of the one-lane classes.
template <int ENC, bool IL>
__device__ __forceinline__ void d_traceback_rows(const DpProb *__restrict__ probs, DpRes *__restrict__ res, int pi, bool have, int lpp, int rb4_, const DpOpt o_,
                                                 const uint8_t *__restrict__ tb_all, uint32_t *__restrict__ cig, int32_t *__restrict__ retry,
                                                 uint32_t *stage)
{
    constexpr bool il = IL, tag8 = ENC == 1, nib = ENC == 2;
    const int rb4 = ENC == 2 ? rb4_ : 0;
    const int lane = threadIdx.x;
    DpProb P = probs[pi];
    if (P.kind >= 3) have = false;
    const int rowb = nib ? rb4 : lpp * 4, dlo = P.dlo, dhi_ = P.dhi, mg = P.pad[0] >> 8;
    int gc = 0;                                                    // tagged spills: cost of the gap runs of the path (piece by piece, as the DP counted them)
    int i = 0, j = 0;
    if (have) { i = res[pi].bi; j = res[pi].bj; }
    const uint8_t *tb = tb_all + P.tb_off;                       // 64-byte aligned (k_prob_sizes)
    uint32_t *cg = cig + P.cig_off;
    int no = 0, ml = 0, mc = 0, state = 0, cur_op = -1, cur_len = 0, touched = 0;
    const int mytop = have && i > 0 && j > 0 ? (i + j) >> 1 : -1;  // highest row this lane reads
    // last byte of row k: rows are rowb bytes apart (nibble rows too: a row pair is 2 * rb4 bytes)
    const int mytopline = mytop >= 0 ? (mytop * rowb + rowb - 1) >> 6 : -1;
    int rtop = mytop;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int v = __shfl_xor(rtop, o); rtop = v > rtop ? v : rtop; }
    rtop = __builtin_amdgcn_readfirstlane(rtop);
    if (rtop >= 0) {
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0, a2 = a0, a3 = a0, b0 = a0, b1 = a0, b2 = a0, b3 = a0;   // even / odd line in flight
        int tag0 = -1, tag1 = -1;                                 // lines in the two LDS slots (wave-uniform)
        const int ltop = (rtop * rowb + rowb - 1) >> 6;
        // il (wave-uniform): the wave's problems are interleaved in 8-byte units (k_tb_gather): line L of this lane = units 8L..8L+7,
        // 512 bytes apart, and each of the eight load instructions reads 512 contiguous bytes for the wave
#define TBR_FETCH(L_, x0, x1, x2, x3) do { const int L__ = (L_); if (L__ >= 0 && L__ <= mytopline) { \
            if (il) { const uint2 *src = (const uint2*)tb + (int64_t)L__ * 512; \
                const uint2 u0 = TB_LD2(src), u1 = TB_LD2(src + 64), u2 = TB_LD2(src + 128), u3 = TB_LD2(src + 192), u4 = TB_LD2(src + 256), u5 = TB_LD2(src + 320), u6 = TB_LD2(src + 384), u7 = TB_LD2(src + 448); \
                x0 = make_uint4(u0.x, u0.y, u1.x, u1.y); x1 = make_uint4(u2.x, u2.y, u3.x, u3.y); x2 = make_uint4(u4.x, u4.y, u5.x, u5.y); x3 = make_uint4(u6.x, u6.y, u7.x, u7.y); } \
            else { const uint4 *src = (const uint4*)(tb + ((int64_t)L__ << 6)); x0 = src[0]; x1 = src[1]; x2 = src[2]; x3 = src[3]; } } } while (0)
#define TBR_PUT(dst_, x0, x1, x2, x3) do { uint32_t *dst = (dst_); \
            dst[0 * 64] = x0.x; dst[1 * 64] = x0.y; dst[2 * 64] = x0.z; dst[3 * 64] = x0.w; dst[4 * 64] = x1.x; dst[5 * 64] = x1.y; dst[6 * 64] = x1.z; dst[7 * 64] = x1.w; \
            dst[8 * 64] = x2.x; dst[9 * 64] = x2.y; dst[10 * 64] = x2.z; dst[11 * 64] = x2.w; dst[12 * 64] = x3.x; dst[13 * 64] = x3.y; dst[14 * 64] = x3.z; dst[15 * 64] = x3.w; } while (0)
        // line L (the next lower one, by construction) moves from its register set into its LDS slot; the set refills with line L-2
#define TBR_ENSURE(L_) do { const int Le = (L_); \
            if (Le & 1) { if (tag1 != Le) { TBR_PUT(stage + 1024 + lane, b0, b1, b2, b3); tag1 = Le; TBR_FETCH(Le - 2, b0, b1, b2, b3); } } \
            else        { if (tag0 != Le) { TBR_PUT(stage + lane, a0, a1, a2, a3); tag0 = Le; TBR_FETCH(Le - 2, a0, a1, a2, a3); } } } while (0)
        if (ltop & 1) { TBR_FETCH(ltop, b0, b1, b2, b3); TBR_FETCH(ltop - 1, a0, a1, a2, a3); }
        else { TBR_FETCH(ltop, a0, a1, a2, a3); TBR_FETCH(ltop - 1, b0, b1, b2, b3); }
        // line by line: with lines L+1 and L in LDS every lane walks on while its next cell lies in one of them, then the
        // wave moves to the next line together (a lane is active in most iterations: it runs ahead inside the two lines
        // instead of waiting for the slowest lane row by row)
        for (int L = ltop; L >= 0; --L) {
            TBR_ENSURE(L);
            const int lim = L << 6;
            for (;;) {
                const int a = i + j, sl = (j - i - dlo) >> 1;
                const int o = nib ? (a >> 1) * rb4 + ((sl >> 2) << 2) + (((sl >> 1) & 1) << 1) + (sl & 1)
                                  : (((a >> 1) * lpp + (sl >> 1)) << 2) + ((a & 1) << 1) + (sl & 1);
                const bool act = i > 0 && j > 0 && o >= lim;
                if (!__any(act)) break;
                const int oo = act ? o : 0;
                // byte of the cell at byte offset x of the problem's trace-back matrix (lines L + 1 and L are in the two LDS slots)
#define TBR_BYTE(x) ((uint32_t)stage8[((((x) >> 6) & 1) * 1024 + (((x) & 63) >> 2) * 64 + lane) * 4 + ((x) & 3)])
                const uint8_t *stage8 = (const uint8_t*)stage;
                uint32_t t = TBR_BYTE(oo);
                // UP TO FOUR DIAGONAL MOVES IN ONE TRIP (round 6): nine path steps in ten are diagonal moves, and the cell one step up
                // the diagonal is the same byte column one row (rowb bytes) up -- for most classes inside the two lines this wave
                // holds in LDS anyway.  The three cells above are read with the current one (independent LDS reads instead of a
                // chain of four round trips), and a run of "H came from the diagonal" cells is ONE trip of the loop.
                const int mij = i < j ? i : j, room = o - lim;
                const bool v1 = act && mij > 1 && room >= rowb, v2 = act && mij > 2 && room >= 2 * rowb, v3 = act && mij > 3 && room >= 3 * rowb;
                const uint32_t t1 = TBR_BYTE(v1 ? oo - rowb : 0), t2 = TBR_BYTE(v2 ? oo - 2 * rowb : 0), t3 = TBR_BYTE(v3 ? oo - 3 * rowb : 0);
#undef TBR_BYTE
                uint32_t nd, mb;                    // bit r: cell r of the diagonal is NOT a diagonal move (or not there); bases equal
                if constexpr (ENC == 2) {                                          // even step: low nibble, odd step: high nibble
                    const int sh = (a & 1) << 2;
                    nd = (uint32_t)(!v1 || ((t1 >> sh) & 3u) != 2u) << 1 | (uint32_t)(!v2 || ((t2 >> sh) & 3u) != 2u) << 2 | (uint32_t)(!v3 || ((t3 >> sh) & 3u) != 2u) << 3;
                    mb = 0;
                    const uint32_t nb = (t >> sh) & 0xfu;                           // raw: tag of H (2 diagonal, 1 E, 0 F), E extended, F opened
                    t = (2u - (nb & 3u)) | (nb & 4u) | ((nb & 8u) ^ 8u);
                } else if constexpr (ENC == 1) {                                    // raw: tag of H (4 - source), E1 / E2 extended, F1 / F2 opened
                    nd = (uint32_t)(!v1 || (t1 & 7u) != 4u) << 1 | (uint32_t)(!v2 || (t2 & 7u) != 4u) << 2 | (uint32_t)(!v3 || (t3 & 7u) != 4u) << 3;
                    mb = 0;
                    t = (4u - (t & 7u)) | (t & 0x28u) | ((t & 0x50u) ^ 0x50u);
                } else {
                    nd = (uint32_t)(!v1 || (t1 & 7u) != 0u) << 1 | (uint32_t)(!v2 || (t2 & 7u) != 0u) << 2 | (uint32_t)(!v3 || (t3 & 7u) != 0u) << 3;
                    mb = (t >> 7) | (t1 >> 7) << 1 | (t2 >> 7) << 2 | (t3 >> 7) << 3;
                }
                if (act) {
                    touched |= (j - i - dlo <= mg) | (dhi_ - (j - i) <= mg);       // within mg diagonals of a band edge
                    const int s0 = state ? state : (int)(t & (nib ? 3 : 7));
                    const int isM = s0 == 0, isD = s0 & 1;
                    const int op = isM ? 0 : (isD ? 2 : 1);
                    // cells of this trip: the run of diagonal moves from the current cell on (1 .. 4), or one step of a gap
                    const int nn = isM ? __builtin_ctz(nd | 16u) : 1;         // (nd bit 0 is clear: the current cell is a diagonal move here)
                    // a step inside a gap run costs its piece's extension, the step that enters the run (walking backwards: the
                    // run's LAST cell) its opening as well
                    if (!isM) gc += (s0 <= 2 ? o_.e : o_.e2) + (state ? 0 : (s0 <= 2 ? o_.q : o_.q2));
                    state = (isM || !((t >> ((nib ? 1 : 2) + s0)) & 1)) ? 0 : s0;
                    ml += isM ? __builtin_popcount(mb & ((1u << nn) - 1u)) : 0; mc += isM ? nn : 0;
                    i -= isM ? nn : (isD ^ 1); j -= isM ? nn : isD;
                    const bool same = op == cur_op;
                    if (!same && cur_len) cg[no++] = (uint32_t)cur_len << 4 | (uint32_t)cur_op;
                    cur_len = same ? cur_len + nn : nn; cur_op = op;
                }
            }
            if (!__any(i > 0 && j > 0)) break;
        }
#undef TBR_FETCH
#undef TBR_PUT
#undef TBR_ENSURE
    }
    if (!have) return;
    // what is left of i or j is the boundary run of row 0 / column 0: the cheaper of the two pieces, as the DP fills it
    if (i > 0) { const int c1 = o_.q + o_.e * i, c2 = o_.q2 + o_.e2 * i; gc += c1 < c2 ? c1 : c2; }
    if (j > 0) { const int c1 = o_.q + o_.e * j, c2 = o_.q2 + o_.e2 * j; gc += c1 < c2 ? c1 : c2; }
    if (i > 0) { if (cur_op == 1) cur_len += i; else { if (cur_len) cg[no++] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; cur_op = 1; cur_len = i; } }
    if (j > 0) { if (cur_op == 2) cur_len += j; else { if (cur_len) cg[no++] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; cur_op = 2; cur_len = j; } }
    if (cur_len) cg[no++] = (uint32_t)cur_len << 4 | (uint32_t)cur_op;
    if (nib || tag8) ml = (res[pi].score + o_.b * mc + gc) / (o_.a + o_.b);      // a ml - b (mc - ml) - gc = score
    res[pi].nops = no; res[pi].mlen = ml; res[pi].mcols = mc;
    if (retry && P.kind == 0 && touched && P.m + P.n <= ADAPT_MAX_STEPS) retry[pi] = 1;
}
// trace-back of the packed fill classes, driven by the same cost-ordered wave table as the forward launch (one block
// per table entry, one lane per problem of that entry): the long problems start first and no class waits for another
#ifdef TB_PROF        /* debug tap (-DTB_PROF): per launch, the waves' own durations against the launch's span -- is the launch its longest wave? */
__device__ unsigned long long g_tb_prof[8];      // [0] sum of wave ticks (100 MHz), [1] max (ticks << 16 | lines), [2] waves, [3] sum of lines, [4] first start, [5] last end
#endif
__global__ void __launch_bounds__(64) k_traceback_pk(const DpProb *__restrict__ probs, DpRes *__restrict__ res,
                                                     const uint8_t *__restrict__ tb_all, uint32_t *__restrict__ cig, int32_t *__restrict__ retry,
                                                     const uint32_t *__restrict__ waves, const int32_t *__restrict__ cls_list, ClsOff off, int32_t tb4, DpOpt o, int32_t tag8_steps)
{
    __shared__ uint32_t stage[TB_SLOTS * 16 * 64];
    const uint32_t w = waves[blockIdx.x];
    const int cls = (int)(w >> 26), first = (int)(w & 0x3ffffffu);
    const int ppw = 64 / PK_LPP[PK_IDX(cls)], t = threadIdx.x;
    const bool have = t < ppw && first + t < off.off[cls + 1] - off.off[cls];
    const int rb4 = d_tb4(cls, tb4) ? d_tb4_rowb(cls) : 0;
    const DpProb P0 = probs[cls_list[off.off[cls] + first]];                   // the wave's longest problem: k_dp_pk chose the cell by it
    const bool tag8 = !rb4 && !(cls == 17 && d_onep_d(o.q, o.e, o.q2, o.e2) >= 16) && !(cls == 10 && d_onep_d(o.q, o.e, o.q2, o.e2) >= 20) && P0.m + P0.n <= tag8_steps;
#ifdef TB_PROF
    const unsigned long long t0_ = wall_clock64();
#endif
    const int pi_ = cls_list[off.off[cls] + (have ? first + t : first)], lpp_ = PK_LPP[PK_IDX(cls)] * PK_R[PK_IDX(cls)];
    if (d_tb_interleaved(cls)) {
        if (rb4) d_traceback_rows<2, true>(probs, res, pi_, have, lpp_, rb4, o, tb_all, cig, retry, stage);
        else if (tag8) d_traceback_rows<1, true>(probs, res, pi_, have, lpp_, 0, o, tb_all, cig, retry, stage);
        else d_traceback_rows<0, true>(probs, res, pi_, have, lpp_, 0, o, tb_all, cig, retry, stage);
    } else if (tag8) d_traceback_rows<1, false>(probs, res, pi_, have, lpp_, 0, o, tb_all, cig, retry, stage);
    else d_traceback_rows<0, false>(probs, res, pi_, have, lpp_, 0, o, tb_all, cig, retry, stage);
#ifdef TB_PROF
    if (threadIdx.x == 0) {
        const unsigned long long t1_ = wall_clock64(), dt = t1_ - t0_;
        const int rowb_ = rb4 ? rb4 : PK_LPP[PK_IDX(cls)] * PK_R[PK_IDX(cls)] * 4;
        unsigned long long lines = (unsigned long long)(((P0.m + P0.n) / 2 + 1) * rowb_ + 63) >> 6; if (lines > 65535) lines = 65535;
        atomicAdd(&g_tb_prof[0], dt); atomicMax(&g_tb_prof[1], dt << 16 | lines); atomicAdd(&g_tb_prof[2], 1ULL); atomicAdd(&g_tb_prof[3], lines);
        atomicMin(&g_tb_prof[4], t0_); atomicMax(&g_tb_prof[5], t1_);
    }
#endif
}

// Trace-back of the few long / wide problems: one WAVE per problem.  Every lane runs the same walk (uniform control
// flow, lane 0 writes); what the other lanes add is memory parallelism: on a miss the wave loads, in one go, the 64-byte
// pieces of the next 64 rows around the current diagonal (the path moves at most one slot per step) into LDS, so the
// walk itself only ever waits for LDS.
__global__ void __launch_bounds__(64) k_traceback_w(const DpProb *__restrict__ probs, DpRes *__restrict__ res, int32_t np,
                                                    const uint8_t *__restrict__ tb_all, uint32_t *__restrict__ cig, int32_t *__restrict__ retry,
                                                    const int32_t *__restrict__ list)
{
    __shared__ uint32_t win[64 * 16];          // window line k (row r0 - k): win[k*16 .. k*16+15]
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= np) return;
    const int pi = list ? list[blockIdx.x] : (int)blockIdx.x;
    const DpProb P = probs[pi];
    if (P.kind >= 3) return;
    const int dhi_ = P.dhi; int touched = 0;
    const int cls = P.pad[0] & 0xff, dlo = P.dlo, mg = P.pad[0] >> 8;
    const int D = P.dhi - dlo + 1, stride = (D + 2) / 2;
    const int lpp = cls >= 5 ? d_cls_slots(cls) : 0;
    const bool packed = cls >= 5;
    const int64_t rowb = packed ? (int64_t)lpp * 4 : stride;        // bytes per row of the trace-back matrix
    const uint8_t *tb = tb_all + P.tb_off;
    int i = res[pi].bi, j = res[pi].bj;
    uint32_t *cg = cig + P.cig_off;
    int no = 0, ml = 0, mc = 0, state = 0, cur_op = -1, cur_len = 0;
    int r0 = -1; int64_t c0 = 0;               // window: rows r0 .. r0-63, bytes [c0, c0+64) of each row (c0 multiple of 4)
    while (i > 0 && j > 0) {
        const int a = i + j, sl = (j - i - dlo) >> 1;
        int row; int64_t col;                  // row and byte column of this cell's trace-back byte
        if (cls >= 10) { row = a >> 1; col = ((int64_t)(sl >> 1) << 2) + ((a & 1) << 1) + (sl & 1); }
        else if (packed) { row = a >> 2; col = ((int64_t)sl << 2) + (a & 3); }
        else { row = a; col = sl; }
        const int k = r0 - row;
        if (r0 < 0 || k < 0 || k >= 64 || col < c0 || col >= c0 + 64) {
            __syncthreads();                   // everybody is done reading the old window
            r0 = row; c0 = col - 32; if (c0 < 0) c0 = 0; c0 &= ~3LL;
            const int rk = r0 - lane;
            uint32_t v[16];
#pragma unroll
            for (int x = 0; x < 16; ++x) v[x] = 0;
            if (rk >= 0) {
                // P.tb_off is a multiple of 16 but rows of the byte-per-cell layout are not: dword loads from the
                // enclosing aligned address, shifted into place
                const int64_t byte0 = (int64_t)rk * rowb + c0;
                const uint32_t *src = (const uint32_t*)(tb + (byte0 & ~3LL));
                const int sh = (int)(byte0 & 3) * 8;
                if (sh == 0) {
#pragma unroll
                    for (int x = 0; x < 16; ++x) v[x] = src[x];
                } else {
                    uint32_t prev = src[0];
#pragma unroll
                    for (int x = 0; x < 16; ++x) { const uint32_t nx = src[x + 1]; v[x] = (prev >> sh) | (nx << (32 - sh)); prev = nx; }
                }
            }
#pragma unroll
            for (int x = 0; x < 16; ++x) win[lane * 16 + x] = v[x];
            __syncthreads();
        }
        const int w = (int)(col - c0);
        touched |= (j - i - dlo <= mg) | (dhi_ - (j - i) <= mg);       // within mg diagonals of a band edge
        if (state == 0) {
            // A DIAGONAL RUN IN ONE TRIP (round 6): nine path steps in ten are diagonal moves and the walk took them one LDS round
            // trip at a time on every lane alike.  Lane k now looks at the cell k steps up the diagonal (i - k, j - k) -- the same
            // diagonal slot, so in every layout a byte of the window that is already in LDS -- and a ballot gives the length of
            // the run of "H came from the diagonal" cells from the current cell on: the whole run is one step of the loop.
            const int ik = i - lane, jk = j - lane, ak = a - 2 * lane;
            int rowk; int64_t colk;
            if (cls >= 10) { rowk = ak >> 1; colk = ((int64_t)(sl >> 1) << 2) + ((ak & 1) << 1) + (sl & 1); }
            else if (packed) { rowk = ak >> 2; colk = ((int64_t)sl << 2) + (ak & 3); }
            else { rowk = ak; colk = sl; }
            const int kk = r0 - rowk, wk = (int)(colk - c0);
            const bool inw = ik > 0 && jk > 0 && kk < 64 && wk >= 0 && wk < 64;
            const uint32_t tk = inw ? (win[kk * 16 + (wk >> 2)] >> ((wk & 3) * 8)) & 0xffu : 0xffu;
            const uint64_t stop = ~__ballot(inw && (tk & 7u) == 0u);
            const int run = stop ? __builtin_ctzll(stop) : 64;
            if (run > 0) {
                const uint64_t low = run >= 64 ? ~0ULL : (1ULL << run) - 1;
                ml += __builtin_popcountll(__ballot(inw && (tk >> 7)) & low); mc += run; i -= run; j -= run;
                if (cur_op == 0) cur_len += run;
                else { if (cur_len && lane == 0) cg[no] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; if (cur_len) ++no; cur_op = 0; cur_len = run; }
                continue;
            }
        }
        const uint32_t t = (win[(r0 - row) * 16 + (w >> 2)] >> ((w & 3) * 8)) & 0xffu;
        if (state == 0) state = t & 7;
        int op;
        if (state == 0) { op = 0; ml += (t >> 7) & 1; ++mc; --i; --j; }
        else if (state == 1) { op = 2; if (!(t & 8))  state = 0; --j; }
        else if (state == 2) { op = 1; if (!(t & 16)) state = 0; --i; }
        else if (state == 3) { op = 2; if (!(t & 32)) state = 0; --j; }
        else                 { op = 1; if (!(t & 64)) state = 0; --i; }
        if (op == cur_op) ++cur_len;
        else { if (cur_len && lane == 0) cg[no] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; if (cur_len) ++no; cur_op = op; cur_len = 1; }
    }
    if (lane != 0) return;
    if (i > 0) { if (cur_op == 1) cur_len += i; else { if (cur_len) cg[no++] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; cur_op = 1; cur_len = i; } }
    if (j > 0) { if (cur_op == 2) cur_len += j; else { if (cur_len) cg[no++] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; cur_op = 2; cur_len = j; } }
    if (cur_len) cg[no++] = (uint32_t)cur_len << 4 | (uint32_t)cur_op;
    res[pi].nops = no; res[pi].mlen = ml; res[pi].mcols = mc;
    if (retry && P.kind == 0 && touched && P.m + P.n <= ADAPT_MAX_STEPS) retry[pi] = 1;
}

// ---- sequence-set subset: copy the packed words of the chosen sequences (64-base padded, so whole words) into a new set;
// with `rc` (nullable) sequence i of the new set is the REVERSE COMPLEMENT of its source when rc[i] is set: the contigs of the
// per-locus realignment (S6 maps every window read to the forward and the reverse-complement contig, TELR_te.py:644-646) are
// turned on the device instead of on the host.  Base j of the result = 3 - base (L - 1 - j), an N stays an N (code 0 + mask bit,
// as the packers write it), the padding behind the sequence is zero.
__global__ void __launch_bounds__(256) k_seq_gather(const uint32_t *__restrict__ src2, const uint32_t *__restrict__ srcn,
                                                    const int64_t *__restrict__ src_boff, const int32_t *__restrict__ idx,
                                                    const int64_t *__restrict__ dst_boff, int32_t n, uint32_t *__restrict__ dst2, uint32_t *__restrict__ dstn,
                                                    const uint8_t *__restrict__ rc = nullptr, const int32_t *__restrict__ dst_len = nullptr)
{
    const int i = blockIdx.x;
    if (i >= n) return;
    const int64_t sb = src_boff[idx[i]], db = dst_boff[i], nb = dst_boff[i + 1] - db;      // padded bases, multiples of 64
    const uint32_t *s2 = src2 + sb / 16, *sn = srcn + sb / 32;
    uint32_t *d2 = dst2 + db / 16, *dn = dstn + db / 32;
    if (rc && rc[i]) {
        const int L = dst_len[i];
        for (int64_t w = threadIdx.x; w < nb / 32; w += blockDim.x) {            // 32 bases: two code words, one mask word
            uint32_t c0 = 0, c1 = 0, m = 0;
            for (int x = 0; x < 32; ++x) {
                const int64_t j = w * 32 + x;
                if (j >= L) break;
                const int64_t p = L - 1 - j;
                const uint32_t isn = (sn[p >> 5] >> (int)(p & 31)) & 1u;
                const uint32_t code = isn ? 0u : 3u - ((s2[p >> 4] >> ((int)(p & 15) * 2)) & 3u);
                if (x < 16) c0 |= code << (2 * x); else c1 |= code << (2 * (x - 16));
                m |= isn << x;
            }
            d2[2 * w] = c0; d2[2 * w + 1] = c1; dn[w] = m;
        }
        return;
    }
    for (int64_t w = threadIdx.x; w < nb / 16; w += blockDim.x) d2[w] = s2[w];
    for (int64_t w = threadIdx.x; w < nb / 32; w += blockDim.x) dn[w] = sn[w];
}

// ---- per-chain numbers from the per-problem results: one wave per kept chain sums its problems (score, matching
// bases, block length) and keeps the reach of the two end extensions; the per-problem records stay on the device.
struct StitchRec { int32_t p0, p1, has_left, pad; };     // problems [p0, p1) of a kept chain
struct ChainStat { int32_t dp, mlen, blen, l_bi, l_bj, r_bi, r_bj, pad; };     // 32 B
__global__ void __launch_bounds__(64) k_chain_stats(const StitchRec *__restrict__ sv, int32_t nk, const DpRes *__restrict__ res, ChainStat *__restrict__ out, int32_t cx_scale)
{
    const int x = blockIdx.x, lane = threadIdx.x;
    if (x >= nk) return;
    const StitchRec S = sv[x];
    int dp = 0, ml = 0, bl = 0;
    for (int z = S.p0 + lane; z < S.p1; z += 64) { const DpRes d = res[z]; dp += d.score; ml += d.mlen; bl += d.bi + d.bj - d.mcols; }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { dp += __shfl_xor(dp, s); ml += __shfl_xor(ml, s); bl += __shfl_xor(bl, s); }
    if (lane == 0) {
        if (cx_scale) { const int v = dp + cx_scale / 2; dp = v >= 0 ? v / cx_scale : -((-v + cx_scale - 1) / cx_scale); }      // convex cost: segment scores are in 1/cx_scale units
        ChainStat c; c.dp = dp; c.mlen = ml; c.blen = bl; c.pad = 0;
        const DpRes a = res[S.p0], b = res[S.p1 - 1];
        c.l_bi = a.bi; c.l_bj = a.bj; c.r_bi = b.bi; c.r_bj = b.bj;
        out[x] = c;
    }
}
// per DP class: problems, cells, steps, algorithmic bytes (telr_last_dp_classes); acc[DP_NCLS*4] = sum of target window bases
__global__ void __launch_bounds__(256) k_dp_account(const DpProb *__restrict__ probs, const DpRes *__restrict__ res, int32_t np, unsigned long long *__restrict__ acc)
{
    __shared__ unsigned long long lacc[DP_NCLS * 4 + 1];
    for (int z = threadIdx.x; z <= DP_NCLS * 4; z += blockDim.x) lacc[z] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) {
        const DpProb P = probs[i]; const DpRes d = res[i];
        const int cls = P.kind >= 3 ? 0 : (P.pad[0] & 0xff);
        atomicAdd(&lacc[cls * 4 + 0], 1ULL);
        atomicAdd(&lacc[cls * 4 + 1], (unsigned long long)d.cells);
        atomicAdd(&lacc[cls * 4 + 2], (unsigned long long)(d.bi + d.bj));
        atomicAdd(&lacc[cls * 4 + 3], (unsigned long long)((d.bi + d.tbases + 3) / 4 + 4 * (long long)d.nops + 32));
        atomicAdd(&lacc[DP_NCLS * 4], (unsigned long long)d.tbases);
    }
    __syncthreads();
    for (int z = threadIdx.x; z <= DP_NCLS * 4; z += blockDim.x) if (lacc[z]) atomicAdd(&acc[z], lacc[z]);
}

// ---- CIGAR stitching of the surviving records on the device: one thread per record walks the raw CIGARs of
// its problems (left extension in emission order, fills and right extension reversed) and merges equal ops
// across problem boundaries.  Pass 1 counts the final ops (they only merge at boundaries), pass 2 writes them.
struct StitchProb { int32_t sv, off, skip, extra; };      // per problem: record, offset inside the record, first op merged away, length absorbed by its last op
// One wave per chain, 64 problems per iteration: the per-problem loads run in parallel, the three things that depend on
// earlier problems (op code of the previous non-empty problem, running op count, the problem whose last op absorbs merged
// first ops) are carried through wave scans.
__device__ __forceinline__ int d_wave_last_valid(int v, bool valid, int carry, int lane)     // value of the nearest lane < this one with valid, else carry
{
    // inclusive "last valid" scan, then shifted by one lane
    int x = valid ? v : INT32_MIN;                       // INT32_MIN = none
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o && x == INT32_MIN) x = y; }
    int prev = __shfl_up(x, 1);
    if (lane == 0 || prev == INT32_MIN) prev = (lane == 0) ? carry : (prev == INT32_MIN ? carry : prev);
    return prev;
}
__global__ void __launch_bounds__(64) k_stitch_count(const StitchRec *__restrict__ sv, int32_t ns, const DpProb *__restrict__ probs, const DpRes *__restrict__ res,
                                                     const uint32_t *__restrict__ raw, int64_t *__restrict__ nfin, StitchProb *__restrict__ sp)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= ns) return;
    const StitchRec S = sv[i];
    int n = 0, prev_code = -1, owner = -1;               // carried across iterations (uniform)
    for (int p0 = S.p0; p0 < S.p1; p0 += 64) {
        const int p = p0 + lane;
        const bool in = p < S.p1;
        int no = 0, fcode = -2, lcode = -3, flen = 0;
        if (in) {
            no = res[p].nops;
            if (no) {
                const int64_t off = probs[p].cig_off;
                const bool fwd = p == S.p0 && S.has_left;
                const uint32_t fo = raw[off + (fwd ? 0 : no - 1)], lo = raw[off + (fwd ? no - 1 : 0)];
                fcode = (int)(fo & 0xf); flen = (int)(fo >> 4); lcode = (int)(lo & 0xf);
            }
        }
        const bool ne = no > 0;
        const int prev = d_wave_last_valid(lcode, ne, prev_code, lane);
        const int skip = ne && prev == fcode ? 1 : 0;
        const int written = ne ? no - skip : 0;
        int inc = written;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(inc, o); if (lane >= o) inc += y; }
        const bool isowner = written > 0;
        const int own_prev = d_wave_last_valid(p, isowner, owner, lane);       // problem whose tail op absorbs this one's merged first op
        if (in) { StitchProb q; q.sv = i; q.off = n + inc - written; q.skip = skip; q.extra = 0; sp[p] = q; }
        __threadfence_block();
        if (skip && own_prev >= 0) atomicAdd(&sp[own_prev].extra, flen);
        // carries: last non-empty problem's last op code, op count, last owner
        const uint64_t nem = __ballot(ne), owm = __ballot(isowner);
        if (nem) prev_code = __shfl(lcode, 63 - __clzll((long long)nem));
        if (owm) owner = p0 + (63 - __clzll((long long)owm));
        n += __shfl(inc, 63);
    }
    if (lane == 0) nfin[i] = n;
}
__global__ void k_stitch_write(int32_t np, const StitchProb *__restrict__ sp, const DpProb *__restrict__ probs, const DpRes *__restrict__ res,
                               const StitchRec *__restrict__ sv, const uint32_t *__restrict__ raw, const int64_t *__restrict__ fin_off, uint32_t *__restrict__ out)
{
    // eight lanes per problem (a problem has ~23 ops): 32-byte pieces of the raw run and of the output per group, and the
    // outputs of neighbouring problems are neighbours
    const int g = blockIdx.x * blockDim.x + threadIdx.x, p = g >> 3, l = g & 7;
    if (p >= np) return;
    const StitchProb q = sp[p];
    if (q.sv < 0) return;
    const int no = res[p].nops;
    if (no - q.skip <= 0) return;
    const uint32_t *src = raw + probs[p].cig_off;
    const bool fwd = p == sv[q.sv].p0 && sv[q.sv].has_left;
    uint32_t *o = out + fin_off[q.sv] + q.off;
    for (int z = q.skip + l; z < no; z += 8) {
        uint32_t op = src[fwd ? z : no - 1 - z];
        if (z == no - 1) op += (uint32_t)q.extra << 4;
        o[z - q.skip] = op;
    }
}

// ---------------------------------------------------------------------------------------
// 6. depth medians (samtools depth -aa -r | statistics.median)
struct DepthRec { int32_t tid, ts, n_cigar, pad; int64_t cigar_off; };
__global__ void k_depth_diff(const DepthRec *__restrict__ recs, int32_t nrec, const uint32_t *__restrict__ cig,
                             const int64_t *__restrict__ toff, int32_t *__restrict__ diff)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrec) return;
    const DepthRec R = recs[r];
    int32_t *d = diff + toff[R.tid];
    int t = R.ts;
    for (int z = 0; z < R.n_cigar; ++z) {
        uint32_t c = cig[R.cigar_off + z]; int op = c & 0xf, l = c >> 4;
        if (op == 0) { atomicAdd(&d[t], 1); atomicAdd(&d[t + l], -1); t += l; }
        else if (op == 2) t += l;
    }
}
#define DEPTH_CAP 8000
__global__ void __launch_bounds__(256) k_depth_median(const int32_t *__restrict__ depth, const int64_t *__restrict__ toff, const int32_t *__restrict__ tlen,
                                                      int32_t n_iv, const int32_t *__restrict__ iv_tid, const int32_t *__restrict__ iv_s,
                                                      const int32_t *__restrict__ iv_e, double *__restrict__ out)
{
    __shared__ int32_t hist[DEPTH_CAP + 1];
    const int v = blockIdx.x;
    if (v >= n_iv) return;
    const int tid = iv_tid[v], L = tlen[tid];
    int s = iv_s[v], e = iv_e[v];
    if (s < 0) s = 0;
    if (e > L - 1) e = L - 1;
    const int n = e - s + 1;
    for (int x = threadIdx.x; x <= DEPTH_CAP; x += blockDim.x) hist[x] = 0;
    __syncthreads();
    const int32_t *d = depth + toff[tid];
    for (int x = threadIdx.x; x < n; x += blockDim.x) { int c = d[s + x]; c = c > DEPTH_CAP ? DEPTH_CAP : c; atomicAdd(&hist[c], 1); }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (n <= 0) { out[v] = __longlong_as_double(0x7ff8000000000000LL); return; }
        // order statistics k1=(n-1)/2, k2=n/2 (0-based) from the histogram
        int k1 = (n - 1) / 2, k2 = n / 2, acc = 0, v1 = -1, v2 = -1;
        for (int x = 0; x <= DEPTH_CAP; ++x) {
            acc += hist[x];
            if (v1 < 0 && acc > k1) v1 = x;
            if (v2 < 0 && acc > k2) { v2 = x; break; }
        }
        out[v] = ((double)v1 + (double)v2) / 2.0;
    }
}
