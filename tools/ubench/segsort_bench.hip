// tools/ubench/segsort_bench.hip -- the segmented LDS sort of telr_amd/csrc/segsort.hip.h alone on the device: segments drawn
// like the anchor lists of configs[2] (gamma-distributed read lengths, ~0.137 anchors per base; keys = strand | position |
// query position | span), every tier checked against std::sort, timed per tier with HIP events.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/ubench/segsort_bench tools/ubench/segsort_bench.hip
// usage: segsort_bench [segments] [mean keys per segment] [seed]
#include "../../telr_amd/csrc/segsort.hip.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const int nseg = argc > 1 ? atoi(argv[1]) : 115000;
    const double mean = argc > 2 ? atof(argv[2]) : 1200.0;
    const unsigned seed = argc > 3 ? atoi(argv[3]) : 1;
    std::mt19937_64 rng(seed);
    std::gamma_distribution<double> g(1.6, mean / 1.6);
    std::vector<int32_t> beg(nseg + 1, 0);
    for (int s = 0; s < nseg; ++s) {
        int n = (int)g(rng);
        if (s % 1000 == 0) n = (int)(rng() % 5);                          // tiny and empty segments
        if (s % 5000 == 1) n = 19000 + (int)(rng() % (SEGSORT_CAP - 19000 + 1));     // the largest tier, up to the cap exactly
        if (n > SEGSORT_CAP) n = SEGSORT_CAP;
        beg[s + 1] = beg[s] + n;
    }
    const size_t nk = beg[nseg];
    std::vector<uint64_t> keys(nk);
    for (int s = 0; s < nseg; ++s) {
        const int n = beg[s + 1] - beg[s];
        for (int i = 0; i < n; ++i) {
            // unique inside the segment: query position i (as seeding emits them), random strand / position; some share the high word
            const uint64_t pos = (i % 97 == 0 && i) ? (keys[beg[s] + i - 1] >> 32) & 0x7fffffff : rng() % 137000000ULL;
            keys[beg[s] + i] = (rng() & 1) << 63 | pos << 32 | (uint64_t)i << 8 | 15;
        }
    }
    uint64_t *d_in, *d_out; int32_t *d_beg, *d_cnt, *d_list;
    CK(hipMalloc(&d_in, nk * 8 + 8)); CK(hipMalloc(&d_out, nk * 8 + 8)); CK(hipMalloc(&d_beg, (nseg + 1) * 4));
    CK(hipMalloc(&d_cnt, (SEGSORT_TIERS + 1) * 4)); CK(hipMalloc(&d_list, (size_t)SEGSORT_TIERS * nseg * 4));
    CK(hipMemcpy(d_in, keys.data(), nk * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_beg, beg.data(), (nseg + 1) * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)k_segsort<1024, 8, LoadKeys>, hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * 8 * 8));
    CK(hipFuncSetAttribute((const void*)k_segsort<1024, 20, LoadKeys>, hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * 20 * 8));
    SegSortArgs A; A.in = d_in; A.out = d_out; A.seg_beg = d_beg; A.seg_end = d_beg + 1; A.src_beg = nullptr; A.order = nullptr; A.nseg = nseg;
    A.tier_cnt = d_cnt; A.tier_list = d_list; A.fb_beg = A.fb_end = nullptr;
    hipEvent_t ev[SEGSORT_TIERS + 2]; for (auto &e : ev) CK(hipEventCreate(&e));
    auto grid = [&](int resident) { return dim3((unsigned)std::min<int64_t>(nseg, (int64_t)256 * resident * 8)); };
    float best[SEGSORT_TIERS + 1]; for (auto &b : best) b = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(d_cnt, 0, (SEGSORT_TIERS + 1) * 4));
        CK(hipEventRecord(ev[0]));
        hipLaunchKernelGGL(k_segsort_classify, dim3((nseg + 255) / 256), dim3(256), 0, 0, A);
        CK(hipEventRecord(ev[1]));
        hipLaunchKernelGGL((k_segsort<1024, 20, LoadKeys>), grid(1), dim3(1024), 1024 * 20 * 8, 0, A, 6, LoadKeys()); CK(hipEventRecord(ev[2]));
        hipLaunchKernelGGL((k_segsort<1024, 8, LoadKeys>), grid(2), dim3(1024), 1024 * 8 * 8, 0, A, 5, LoadKeys()); CK(hipEventRecord(ev[3]));
        hipLaunchKernelGGL((k_segsort<512, 8, LoadKeys>), grid(4), dim3(512), 512 * 8 * 8, 0, A, 4, LoadKeys()); CK(hipEventRecord(ev[4]));
        hipLaunchKernelGGL((k_segsort<256, 8, LoadKeys>), grid(8), dim3(256), 256 * 8 * 8, 0, A, 3, LoadKeys()); CK(hipEventRecord(ev[5]));
        hipLaunchKernelGGL((k_segsort<128, 8, LoadKeys>), grid(16), dim3(128), 128 * 8 * 8, 0, A, 2, LoadKeys()); CK(hipEventRecord(ev[6]));
        hipLaunchKernelGGL((k_segsort<64, 8, LoadKeys>), grid(32), dim3(64), 64 * 8 * 8, 0, A, 1, LoadKeys()); CK(hipEventRecord(ev[7]));
        hipLaunchKernelGGL((k_segsort<64, 2, LoadKeys>), grid(32), dim3(64), 64 * 2 * 8, 0, A, 0, LoadKeys()); CK(hipEventRecord(ev[8]));
        CK(hipDeviceSynchronize());
        float tot = 0;
        for (int t = 0; t <= SEGSORT_TIERS; ++t) { float ms; CK(hipEventElapsedTime(&ms, ev[t], ev[t + 1])); best[t] = std::min(best[t], ms); tot += ms; }
        if (rep == 4) printf("last repetition: %.3f ms in all\n", tot);
    }
    std::vector<uint64_t> out(nk); CK(hipMemcpy(out.data(), d_out, nk * 8, hipMemcpyDeviceToHost));
    int32_t cnt[SEGSORT_TIERS + 1]; CK(hipMemcpy(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost));
    size_t bad = 0; size_t tier_keys[SEGSORT_TIERS + 1] = {0};
    for (int s = 0; s < nseg; ++s) {
        const int n = beg[s + 1] - beg[s];
        if (n > 0) tier_keys[segsort_tier_of(n)] += n;
        std::sort(keys.begin() + beg[s], keys.begin() + beg[s + 1]);
        for (int i = 0; i < n; ++i) if (out[beg[s] + i] != keys[beg[s] + i]) { if (bad < 5) fprintf(stderr, "segment %d (n = %d) differs at %d\n", s, n, i); ++bad; break; }
    }
    const char *names[] = { "classify", "<1024,20>", "<1024,8>", "<512,8>", "<256,8>", "<128,8>", "<64,8>", "<64,2>" };
    const int tier_of_launch[] = { -1, 6, 5, 4, 3, 2, 1, 0 };
    float sum = 0;
    for (int t = 0; t <= SEGSORT_TIERS; ++t) {
        sum += best[t];
        if (t == 0) printf("%-10s %8.3f ms\n", names[t], best[t]);
        else { const int tr = tier_of_launch[t]; printf("%-10s %8.3f ms  %8d segments %11zu keys  %7.1f G keys/s\n", names[t], best[t], cnt[tr], tier_keys[tr], tier_keys[tr] / best[t] / 1e6); }
    }
    printf("%d segments, %zu keys: %.3f ms (best of 5 per launch) = %.1f G keys/s, %.2f TB/s of read + write; %zu segments wrong\n", nseg, nk, sum, nk / sum / 1e6, nk * 16 / sum / 1e9, bad);
    return bad ? 1 : 0;
}
