// telr_engine.hip — host runtime and C ABI (include/telr_hip.h) of the MI355X alignment
// engine.  Pipeline of one telr_map() batch, all on one HIP stream:
//   sketch -> seed (count/scan/fill) -> per-query radix sort -> chain DP -> peaks ->
//   back-track -> [D2H chain boxes, host chain selection] -> DP problem list ->
//   banded DP + trace-back -> cigar gather -> [D2H, host assembly + mapq].
// Host steps carry the -N / -p / -M / --secondary semantics and never
// touch bases; every base-level operation runs on the device.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <atomic>
#include <immintrin.h>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <sys/mman.h>
#include <vector>
#include "../../include/telr_hip.h"
#include "kernels.hip.h"
#define TELR_HAVE_SEED_ARGS 1
#include "segsort.hip.h"
#include "radix.hip.h"

// ---------------------------------------------------------------------------------------
static const char *STAGE_NAMES[TELR_N_STAGES] = {
    "sketch", "seed", "sort", "chain", "backtrack", "select_host", "segments", "dp", "stitch_d2h", "d2h", "assemble_host", "index_build",
    "k_dp_reg", "map_wall", "k_traceback", "k_dp_pk"
};
enum { ST_SKETCH, ST_SEED, ST_SORT, ST_CHAIN, ST_BACKTRACK, ST_SELECT, ST_SEGMENTS, ST_DP, ST_GATHER, ST_D2H, ST_ASSEMBLE, ST_INDEX,
       ST_K_REG, ST_MAP_WALL, ST_K_TRACEBACK, ST_K_PK };

#define TELR_NSIDE 8
struct DBuf { void *p = nullptr; size_t bytes = 0; };
struct TileList { std::vector<int32_t> seq, u0, first; int32_t n = 0; };     // tiles of the sketch kernel over a range of sequences

struct telr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side[8] = {nullptr};
    hipEvent_t ev_fork = nullptr, ev_side[8] = {nullptr}, ev_chunk[8] = {nullptr};
    hipStream_t tb_stream = nullptr;      // packed trace-back chunks run here, underneath the next forward chunk
    hipStream_t copy_stream = nullptr;    // the result CIGAR DMA: may still run when telr_map has returned and the next call starts
    hipEvent_t ev_stitched = nullptr, ev_dma = nullptr; bool dma_inflight = false;
    std::string err;
    std::map<std::string, DBuf> bufs;     // grow-only device scratch, reused across calls
    std::map<std::string, DBuf> hbufs;    // grow-only pinned host staging buffers
    std::vector<std::pair<uint32_t*, size_t>> cig_pool;   // recycled result CIGAR buffers
    std::vector<std::vector<telr_aln>> aln_pool;          // recycled result record arrays (a fresh 50-MB vector is page-faulted on every call)
    // grow-only host scratch of map_batch / map_range (same reason: no allocation, no first-touch faults in the steady state)
    std::vector<telr_aln> h_kal, h_stage;
    std::vector<int32_t> h_nsurv, h_poffv;
    TileList tiles;                                       // a batch of 1 Gbp has ~1 M tiles
    int debug = 0;                        // keep stage-level captures for the parity tests
    float stage_ms[TELR_N_STAGES] = {0};
    telr_counters ctr = {};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_st[TELR_N_STAGES][2] = {{nullptr}}; uint32_t st_pending = 0;
    hipEvent_t evk[6] = {nullptr};        // un-synchronised markers around single kernels
    int64_t dpcls[TELR_N_DPCLS * 4] = {0};
    int64_t dp_retries = 0;
    int64_t pk_launches = 0;              // k_dp_pk launches of first DP passes in the last telr_map call (one per range)
    bool background = false;              // streams at the lowest priority (telr_init_background)
    telr_ctx *twin_owner = nullptr; int64_t twin_bases = 0;       // set for the duration of a telr_map call on the contexts that append to its result
    uint32_t *twin_pool = nullptr; size_t twin_pool_cap = 0;      // device CIGAR array of a freed result (TELR_MF_KEEP_CIGARS), for the next call
    struct BamSink *bam_sink = nullptr;   // an output file being prepared for telr_write_bam_dev (bam_dev.hip.h)
    telr_ctx *slot1 = nullptr;            // the second range slot (a context of the same kind: own streams and scratch)
    bool pipe_nomem = false;              // two ranges in flight once ran out of device memory: later calls run one at a time
    char devname[256] = {0};
    int n_cu = 0;                    // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    // debug captures of the last batch (device pointers stay valid until the next call)
    int64_t dbg_na = 0; int32_t dbg_nq = 0;
    std::vector<int32_t> dbg_chain;       // 9 ints per chain
};

#define HIPCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    ctx->err = std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + ":" + std::to_string(__LINE__); \
    return _e == hipErrorOutOfMemory ? TELR_E_NOMEM : TELR_E_HIP; } } while (0)
#define TRY(expr) do { int _r = (expr); if (_r != TELR_OK) return _r; } while (0)
#define TELR_SPLIT_RANGE (-100)      // internal: a batch with >= 2^31 anchors; map_range() halves it

// ---- environment: TWO switches carry every alternative form and every trace the engine still has (read once per process) ----------
//   TELR_AB=tok[,tok...]     A/B forms of earlier rounds, kept because tests/test_gpu_switches.py holds each of them to the same bits:
//                            sort64 (rocPRIM's segmented sort for every query), seed_unfused (seeding and sorting as two kernels),
//                            chain_push (every link of the look-back scored), sketch64 (the 64-bit sketch kernel for k <= 15 too), mz_compact (compacted minimizer arrays), dp_one_wave (the widest LDS classes with one wave per problem), no_islands
//                            (one wave per query in every chaining call), vote_filter (table filter in the vote presets' lookups),
//                            no_pk / no_pkw / no_pkext (int32 classes instead of the packed fills / wide fills / extensions), tb8 (byte
//                            spill in the one-piece classes), no_tag8 (untagged two-piece cell), tb_one_launch (ONE trace-back launch
//                            after all forward kernels), no_avx2 (scalar host packer), fasta_copy (no in-place use of the mapped file),
//                            bam_no_populate, bam_no_twin (BAM writer: no pre-faulted mapping / CIGARs uploaded again)
//   TELR_TRACE=tok[,tok...]  stderr traces: host (wall-clock marks of every batch's host side), mem (device memory at the points
//                            where the engine runs out of it or gives it back), fasta (phases of telr_fasta_load)
// The others are operational: TELR_DEBUG, TELR_HOST_THREADS, TELR_PACK_THREADS, TELR_BATCH_MBP / TELR_BATCH_KBP (range size),
// TELR_PIPELINE (ranges in flight), TELR_SERIAL (profiling: every DP class on the main stream), TELR_TEST_PIPE_NOMEM (test hook).
static bool env_token(const char *var, const char *tok)
{
    const char *e = getenv(var);
    if (!e) return false;
    const size_t n = strlen(tok);
    for (const char *p = e; *p; ) {
        const char *q = strchr(p, ',');
        const size_t len = q ? (size_t)(q - p) : strlen(p);
        if (len == n && !strncmp(p, tok, n)) return true;
        if (!q) break;
        p = q + 1;
    }
    return false;
}
static inline bool ab_on(const char *tok) { return env_token("TELR_AB", tok); }
static inline bool trace_on(const char *tok) { return env_token("TELR_TRACE", tok); }

// TELR_TRACE=mem: device memory at the points where the engine runs out of it or gives it back (stderr)
static void mem_note(telr_ctx *ctx, const char *tag)
{
    static const bool on = trace_on("mem");
    if (!on) return;
    size_t fr = 0, tot = 0; (void)hipMemGetInfo(&fr, &tot);
    size_t own = 0, bam = 0;
    for (auto &kv : ctx->bufs) { own += kv.second.bytes; if (kv.first.compare(0, 4, "bam_") == 0) bam += kv.second.bytes; }
    size_t s1 = 0; if (ctx->slot1) for (auto &kv : ctx->slot1->bufs) s1 += kv.second.bytes;
    fprintf(stderr, "[telr mem] %-44s free %.1f of %.1f GB; this context %.1f GB (writer %.1f), second slot %.1f GB\n", tag, fr / 1e9, tot / 1e9, own / 1e9, bam / 1e9, s1 / 1e9);
}
static int ctx_buf(telr_ctx *ctx, const char *name, size_t bytes, void **out)
{
    DBuf &b = ctx->bufs[name];
    if (b.bytes < bytes || !b.p) {
        if (b.p) HIPCHK(hipFree(b.p));
        b.p = nullptr; b.bytes = 0;
        size_t want = bytes + bytes / 8 + 256;
        HIPCHK(hipMalloc(&b.p, want));
        b.bytes = want;
    }
    *out = b.p;
    return TELR_OK;
}
static int ctx_hbuf(telr_ctx *ctx, const char *name, size_t bytes, void **out)
{
    DBuf &b = ctx->hbufs[name];
    if (b.bytes < bytes || !b.p) {
        if (b.p) HIPCHK(hipHostFree(b.p));
        b.p = nullptr; b.bytes = 0;
        size_t want = bytes + bytes / 8 + 256;
        HIPCHK(hipHostMalloc(&b.p, want, hipHostMallocDefault));
        b.bytes = want;
    }
    *out = b.p;
    return TELR_OK;
}
template <typename T> static int ctx_hbuf_t(telr_ctx *ctx, const char *name, size_t n, T **out)
{
    void *p; TRY(ctx_hbuf(ctx, name, (n ? n : 1) * sizeof(T), &p)); *out = (T*)p; return TELR_OK;
}
template <typename T> static int ctx_buf_t(telr_ctx *ctx, const char *name, size_t n, T **out)
{
    void *p; TRY(ctx_buf(ctx, name, (n ? n : 1) * sizeof(T), &p)); *out = (T*)p; return TELR_OK;
}

// ---------------------------------------------------------------------------------------
// Segmented sort of 64-bit keys (segsort.hip.h): every segment of at most SEGSORT_CAP keys is sorted by ONE workgroup in
// LDS; `any_over` (known to the caller from the anchor counts) sends the larger ones through rocPRIM afterwards.
// TELR_AB=sort64 keeps the library sort for every segment (A/B).  out[beg[s] .. end[s]) <- sorted in[...]; src_beg (nullable)
// gives the segments' places in `in` when they differ from their places in `out`.  fb_in (round 6): where the OVER-SIZE segments' keys lie
// (at their places in `out`) when the producer makes the others on the spot or reads them through src_beg -- only those queries take the
// two-step form then, not the whole range.
template <class P = LoadKeys>
static int seg_sort_u64(telr_ctx *ctx, const char *tag, const uint64_t *in, uint64_t *out, const int32_t *beg, const int32_t *end, const int64_t *src_beg,
                        const int32_t *order, int nseg, size_t nkeys, bool any_over, hipStream_t st, P prod = P(), const uint64_t *fb_in = nullptr, hipEvent_t fb_ready = nullptr)
{
    static const bool lib_sort = ab_on("sort64");
    if (nseg <= 0 || nkeys == 0) return TELR_OK;
    if (lib_sort) {
        if (src_beg) return TELR_E_ARG;              // the caller compacts first
        size_t tb = 0;
        // (all 64 bits: rocprim's segmented sort mis-orders keys with bit 63 set when begin_bit > 0 -- measured, ROCm 7.2)
        HIPCHK(rocprim::segmented_radix_sort_keys(nullptr, tb, in, out, (unsigned)nkeys, (unsigned)nseg, beg, end, 0, 64, st));
        void *tmp; TRY(ctx_buf(ctx, "rp_tmp", tb, &tmp));
        HIPCHK(rocprim::segmented_radix_sort_keys(tmp, tb, in, out, (unsigned)nkeys, (unsigned)nseg, beg, end, 0, 64, st));
        return TELR_OK;
    }
    if (src_beg && any_over && !fb_in) return TELR_E_ARG;
    // the opt-in to more than 64 KiB of dynamic LDS belongs to the DEVICE's function object: one bit per device (and per producer
    // type: a function-local static of the template instance); setting it twice from two host threads is harmless
    static std::atomic<uint64_t> attr_dev{0};
    const uint64_t dev_bit = 1ULL << (ctx->device & 63);
    if (!(attr_dev.load(std::memory_order_acquire) & dev_bit)) {
        HIPCHK(hipFuncSetAttribute((const void*)k_segsort<1024, 8, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * 8 * 8));
        HIPCHK(hipFuncSetAttribute((const void*)k_segsort<1024, 20, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * 20 * 8));
        attr_dev.fetch_or(dev_bit, std::memory_order_release);
    }
    SegSortArgs A; A.in = in; A.out = out; A.seg_beg = beg; A.seg_end = end; A.src_beg = src_beg; A.order = order; A.nseg = nseg;
    const std::string T(tag);
    TRY(ctx_buf_t(ctx, (T + "_tcnt").c_str(), SEGSORT_TIERS + 1, &A.tier_cnt));
    TRY(ctx_buf_t(ctx, (T + "_tlist").c_str(), (size_t)SEGSORT_TIERS * nseg, &A.tier_list));
    A.fb_beg = A.fb_end = nullptr;
    if (any_over) { TRY(ctx_buf_t(ctx, (T + "_fbbeg").c_str(), (size_t)nseg, &A.fb_beg)); TRY(ctx_buf_t(ctx, (T + "_fbend").c_str(), (size_t)nseg, &A.fb_end)); }
    HIPCHK(hipMemsetAsync(A.tier_cnt, 0, (SEGSORT_TIERS + 1) * 4, st));
    hipLaunchKernelGGL(k_segsort_classify, dim3((nseg + 255) / 256), dim3(256), 0, st, A);
    // the library's sort of the over-size segments runs on a side stream UNDER the tiers (a few segments of 10^5 keys: long, narrow
    // launches that the main stream's work would otherwise queue behind); the main stream joins at the end
    hipStream_t fbs = any_over ? ctx->side[1] : nullptr;
    if (any_over) {
        HIPCHK(hipEventRecord(ctx->ev_side[1], st)); HIPCHK(hipStreamWaitEvent(fbs, ctx->ev_side[1], 0));
        if (fb_ready) HIPCHK(hipStreamWaitEvent(fbs, fb_ready, 0));        // the over-size queries' keys are written on another side stream
        const uint64_t *fin = fb_in ? fb_in : in;
        size_t tb = 0;
        HIPCHK(rocprim::segmented_radix_sort_keys(nullptr, tb, fin, out, (unsigned)nkeys, (unsigned)nseg, A.fb_beg, A.fb_end, 0, 64, fbs));
        void *tmp; TRY(ctx_buf(ctx, "rp_tmp_fb", tb, &tmp));
        HIPCHK(rocprim::segmented_radix_sort_keys(tmp, tb, fin, out, (unsigned)nkeys, (unsigned)nseg, A.fb_beg, A.fb_end, 0, 64, fbs));
        HIPCHK(hipEventRecord(ctx->ev_side[1], fbs));
    }
    const int ncu = ctx->n_cu > 0 ? ctx->n_cu : 256;
    auto grid = [&](int resident) { return dim3((unsigned)std::min<int64_t>(nseg, (int64_t)ncu * resident * 8)); };
    // largest tiers first: their few long-running workgroups start while the device is otherwise idle
    hipLaunchKernelGGL((k_segsort<1024, 20, P>), grid(1), dim3(1024), 1024 * 20 * 8, st, A, 6, prod);
    hipLaunchKernelGGL((k_segsort<1024, 8, P>), grid(2), dim3(1024), 1024 * 8 * 8, st, A, 5, prod);
    hipLaunchKernelGGL((k_segsort<512, 8, P>), grid(4), dim3(512), 512 * 8 * 8, st, A, 4, prod);
    hipLaunchKernelGGL((k_segsort<256, 8, P>), grid(8), dim3(256), 256 * 8 * 8, st, A, 3, prod);
    hipLaunchKernelGGL((k_segsort<128, 8, P>), grid(16), dim3(128), 128 * 8 * 8, st, A, 2, prod);
    hipLaunchKernelGGL((k_segsort<64, 8, P>), grid(32), dim3(64), 64 * 8 * 8, st, A, 1, prod);
    hipLaunchKernelGGL((k_segsort<64, 2, P>), grid(32), dim3(64), 64 * 2 * 8, st, A, 0, prod);
    HIPCHK(hipGetLastError());
    if (any_over) HIPCHK(hipStreamWaitEvent(st, ctx->ev_side[1], 0));
    return TELR_OK;
}

// Stage timing.  GPU stages drop a pair of events on the stream and are read back by stage_collect() after the call's
// last synchronisation: timing never drains the stream between stages.
struct StageTimer {
    telr_ctx *ctx; int stage; bool gpu; std::chrono::steady_clock::time_point t0;
    StageTimer(telr_ctx *c, int s, bool g) : ctx(c), stage(s), gpu(g) {
        if (gpu) (void)hipEventRecord(ctx->ev_st[stage][0], ctx->stream); else t0 = std::chrono::steady_clock::now();
    }
    void stop() {
        if (gpu) { (void)hipEventRecord(ctx->ev_st[stage][1], ctx->stream); ctx->st_pending |= 1u << stage; }
        else ctx->stage_ms[stage] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
};
static void stage_collect(telr_ctx *ctx)
{
    for (int z = 0; z < TELR_N_STAGES; ++z) if (ctx->st_pending >> z & 1u) {
        float ms = 0;
        if (hipEventSynchronize(ctx->ev_st[z][1]) == hipSuccess && hipEventElapsedTime(&ms, ctx->ev_st[z][0], ctx->ev_st[z][1]) == hipSuccess) ctx->stage_ms[z] += ms;
    }
    ctx->st_pending = 0;
}

// ---------------------------------------------------------------------------------------
extern "C" const char *telr_strerror(int code)
{
    switch (code) {
    case TELR_OK: return "ok";
    case TELR_E_NODEVICE: return "no usable HIP device";
    case TELR_E_HIP: return "HIP runtime error";
    case TELR_E_ARG: return "invalid argument";
    case TELR_E_RANGE: return "input exceeds the engine's coordinate range (targets < 2^31 bases, queries < 2^24 bases)";
    case TELR_E_NOMEM: return "out of device memory";
    case TELR_E_IO: return "file could not be opened, mapped or written";
    default: return "unknown error";
    }
}
extern "C" const char *telr_last_error(const telr_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }
extern "C" const char *telr_stage_name(int i) { return i >= 0 && i < TELR_N_STAGES ? STAGE_NAMES[i] : ""; }
extern "C" int telr_stage_ms(const telr_ctx *ctx, float *ms) { if (!ctx || !ms) return TELR_E_ARG; memcpy(ms, ctx->stage_ms, sizeof(ctx->stage_ms)); return TELR_OK; }
extern "C" int telr_last_counters(const telr_ctx *ctx, telr_counters *out) { if (!ctx || !out) return TELR_E_ARG; *out = ctx->ctr; return TELR_OK; }
extern "C" int telr_last_dp_classes(const telr_ctx *ctx, int64_t *out) { if (!ctx || !out) return TELR_E_ARG; memcpy(out, ctx->dpcls, sizeof(ctx->dpcls)); return TELR_OK; }

// background = true: every stream of the context (and of its second range slot) is created at the device's lowest priority, so
// that the kernels of another context of the process are dispatched ahead of this one's (telr_init_background)
static int ctx_init(int device, bool background, telr_ctx **out)
{
    if (!out) return TELR_E_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return TELR_E_NODEVICE;
    if (hipSetDevice(device) != hipSuccess) return TELR_E_NODEVICE;
    telr_ctx *ctx = new telr_ctx();
    ctx->device = device; ctx->background = background;
    int prio_least = 0, prio_greatest = 0;
    if (background && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) prio_least = 0;
    auto hipStreamCreate = [&](hipStream_t *st) { return background ? hipStreamCreateWithPriority(st, hipStreamDefault, prio_least) : ::hipStreamCreate(st); };
    if (const char *e = getenv("TELR_DEBUG")) ctx->debug = atoi(e);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) { snprintf(ctx->devname, sizeof(ctx->devname), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount); ctx->n_cu = prop.multiProcessorCount; }
    if (hipStreamCreate(&ctx->stream) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) { delete ctx; return TELR_E_NODEVICE; }
    for (int i = 0; i < 6; ++i) if (hipEventCreate(&ctx->evk[i]) != hipSuccess) { delete ctx; return TELR_E_NODEVICE; }
    for (int i = 0; i < TELR_N_STAGES; ++i) for (int j = 0; j < 2; ++j) if (hipEventCreate(&ctx->ev_st[i][j]) != hipSuccess) { delete ctx; return TELR_E_NODEVICE; }
    for (int i = 0; i < TELR_NSIDE; ++i) if (hipStreamCreate(&ctx->side[i]) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_side[i], hipEventDisableTiming) != hipSuccess) { delete ctx; return TELR_E_NODEVICE; }
    for (int i = 0; i < 8; ++i) if (hipEventCreateWithFlags(&ctx->ev_chunk[i], hipEventDisableTiming) != hipSuccess) { delete ctx; return TELR_E_NODEVICE; }
    if (hipStreamCreate(&ctx->tb_stream) != hipSuccess) { delete ctx; return TELR_E_NODEVICE; }
    if (hipStreamCreate(&ctx->copy_stream) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_stitched, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_dma, hipEventDisableTiming) != hipSuccess) { delete ctx; return TELR_E_NODEVICE; }
    *out = ctx;
    return TELR_OK;
}
extern "C" int telr_init(int device, telr_ctx **out) { return ctx_init(device, false, out); }
extern "C" int telr_init_background(int device, telr_ctx **out) { return ctx_init(device, true, out); }
static void bam_sink_drop(telr_ctx *ctx);
extern "C" void telr_destroy(telr_ctx *ctx)
{
    if (!ctx) return;
    bam_sink_drop(ctx);
    if (ctx->twin_pool) { (void)hipFree(ctx->twin_pool); ctx->twin_pool = nullptr; }
    if (ctx->slot1) { telr_destroy(ctx->slot1); ctx->slot1 = nullptr; }
    (void)hipSetDevice(ctx->device);
    for (auto &kv : ctx->bufs) if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto &kv : ctx->hbufs) if (kv.second.p) (void)hipHostFree(kv.second.p);
    for (auto &pc : ctx->cig_pool) if (pc.first) (void)hipHostFree(pc.first);
    ctx->cig_pool.clear();
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    for (int i = 0; i < TELR_NSIDE; ++i) { if (ctx->side[i]) (void)hipStreamDestroy(ctx->side[i]); if (ctx->ev_side[i]) (void)hipEventDestroy(ctx->ev_side[i]); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (int i = 0; i < 8; ++i) if (ctx->ev_chunk[i]) (void)hipEventDestroy(ctx->ev_chunk[i]);
    if (ctx->tb_stream) (void)hipStreamDestroy(ctx->tb_stream);
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    if (ctx->ev_stitched) (void)hipEventDestroy(ctx->ev_stitched);
    if (ctx->ev_dma) (void)hipEventDestroy(ctx->ev_dma);
    for (int i = 0; i < 6; ++i) if (ctx->evk[i]) (void)hipEventDestroy(ctx->evk[i]);
    for (int i = 0; i < TELR_N_STAGES; ++i) for (int j = 0; j < 2; ++j) if (ctx->ev_st[i][j]) (void)hipEventDestroy(ctx->ev_st[i][j]);
    delete ctx;
}
extern "C" int telr_device_name(const telr_ctx *ctx, char *buf, int buflen)
{
    if (!ctx || !buf || buflen <= 0) return TELR_E_ARG;
    snprintf(buf, buflen, "%s", ctx->devname);
    return TELR_OK;
}

// ---------------------------------------------------------------------------------------
// presets (mirrors telr_amd/presets.py; tests/test_abi.py keeps them equal)
extern "C" int telr_preset(const char *name, telr_idx_opt *io, telr_map_opt *mo)
{
    if (!name || !io || !mo) return TELR_E_ARG;
    std::string s(name);
    io->k = 15; io->w = 10; io->is_hpc = 0; io->bucket_bits = 0;
    memset(mo, 0, sizeof(*mo));
    mo->mid_occ_frac = 2e-4f; mo->min_mid_occ = 10; mo->max_mid_occ = 1000000;
    mo->max_gap = 5000; mo->bw = 500; mo->chain_lookback = 128; mo->min_cnt = 3; mo->min_chain_score = 40;
    mo->mask_level = 0.5f; mo->pri_ratio = 0.8f; mo->best_n = 5; mo->secondary = 1;
    mo->a = 2; mo->b = 4; mo->q = 4; mo->e = 2; mo->q2 = 24; mo->e2 = 1; mo->sc_ambi = 1; mo->zdrop = 400;
    mo->min_dp_max = 80; mo->min_ksw_len = 200; mo->ext_max = 2048; mo->ext_band = 31; mo->flags = TELR_MF_CIGAR; mo->fill_band_q4 = 6; mo->fill_margin = 1;
    if (s == "map-ont") { mo->fill_band_q4 = 4; mo->bw_long = 20000; }                 // -r500,20000: long join
    else if (s == "map-pb") { io->k = 19; io->is_hpc = 1; mo->fill_band_q4 = 12; mo->bw_long = 20000; mo->ext_max = 3900; }      // (ext_max: round 6, the hard genome -- telr_amd/presets.py)
    else if (s == "ngmlr-ont" || s == "ngmlr-pacbio") {
        // `ngmlr -x ont|pacbio` (TELR_alignment.py:28-51, the reference's default aligner).  NGMLR 0.2.7 indexes 13-mers at
        // every third reference position: (w,k) = (5,13) minimizers have that density.  Its convex gap cost (open, then an
        // extension penalty that decays by 0.15 per base from `max` to `min`; Sedlazeck 2018) is the lower envelope of two
        // affine pieces:  pacbio (2, -5, open 5, extend 5 -> 1): C(L) = 5 + sum_{i<L} max(1, 5 - 0.15 i) ~ min(6 + 4L, 60 + L)
        // (exact at L = 1 and L >= 27, within 6 in between);  ont (1, -1, open 1, extend 1 -> 0.5), doubled to integers:
        // C(L) = 2 + sum_{i<L} max(1, 2 - 0.3 i) ~ min(2 + 2L, 4 + L).  AS is in these integer units (ont: twice NGMLR's).
        io->k = 13; io->w = 5;
        // round 4: the convex cost is the spec in EXACT form (cx_*: scores in 1/10 (ont) or 1/20 (pacbio) of the preset's unit; what the
        // next base of a gap costs rides with every gap cell): against the envelope it moved 3.2 % (ont) / 0.18 % (pacbio) of the
        // records' coordinates (faithful gate).  q / e / q2 / e2 stay the envelope's: `cx_scale = 0` over the preset is that A/B.
        if (s == "ngmlr-ont") { mo->a = 2; mo->b = 2; mo->q = 2; mo->e = 2; mo->q2 = 4; mo->e2 = 1; mo->cx_scale = 10; mo->cx_open = 20; mo->cx_ext_max = 20; mo->cx_ext_min = 10; mo->cx_decay = 3; }
        else { mo->a = 2; mo->b = 5; mo->q = 6; mo->e = 4; mo->q2 = 60; mo->e2 = 1; mo->cx_scale = 20; mo->cx_open = 100; mo->cx_ext_max = 100; mo->cx_ext_min = 20; mo->cx_decay = 3; }
        mo->fill_band_q4 = 12; mo->fill_margin = 2;
        if (s == "ngmlr-ont") { mo->fill_band_q4 = 7; mo->fill_margin = 4; mo->ext_band = 63; mo->zdrop = 100; }      // round 5: see telr_amd/presets.py (fills: same results, 18 % fewer cells; extensions: the 0.5 % rule)
        mo->vote_len = 256; mo->vote_bin_shift = 5; mo->vote_min = 3; mo->vote_frac_q8 = 128;      // NGMLR's sub-read voting
        mo->chain_lookback = 256;              // round 6: the 0.5 % rule on the hard genome (telr_amd/presets.py)
    }
    else if (s == "asm10") {
        io->k = 19; io->w = 19; mo->min_mid_occ = 50; mo->max_mid_occ = 500; mo->bw = 10000; mo->max_gap = 10000;
        mo->a = 1; mo->b = 9; mo->q = 16; mo->e = 2; mo->q2 = 41; mo->e2 = 1; mo->min_dp_max = 200; mo->zdrop = 200; mo->best_n = 50;
    } else return TELR_E_ARG;
    mo->chain_gap_q8 = (int32_t)(0.01 * 0.8 * io->k * 256 + 0.5);
    mo->chain_skip_q8 = 0;
    return TELR_OK;
}

// ---------------------------------------------------------------------------------------
// sequence sets
extern "C" void telr_seqset_free(telr_seqset *s);
struct telr_seqset {
    telr_ctx *ctx;
    int32_t n = 0;
    int64_t total_bases = 0, padded_bases = 0;
    std::vector<int64_t> boff;    // [n+1]
    std::vector<int32_t> len;     // [n]
    uint32_t *d_seq2 = nullptr, *d_nmask = nullptr;
    int64_t *d_boff = nullptr; int32_t *d_len = nullptr;
    int32_t max_len = 0;
};

static inline uint8_t nt4_of(unsigned char c)
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': case 'U': case 'u': return 3; default: return 4; }
}

// 8 ASCII bases -> 16 bits of 2-bit codes (A 0, C 1, G 2, T/U 3, either case; base i in bits 2i..2i+1) and 8 ambiguity
// bits (anything else; its code bits are 0), branch-free on one 64-bit word: the packing of a multi-Gbp read set is
// part of the host-inclusive rate of the path
static inline void pack8(uint64_t x, uint32_t &code16, uint32_t &amb8)
{
    const uint64_t L = 0x0101010101010101ULL, H = 0x8080808080808080ULL, M7 = 0x7F7F7F7F7F7F7F7FULL;
    uint64_t t = (x >> 1) & (3 * L);             // bits 1-2 of the ASCII code: A 0, C 1, T 2, G 3
    t ^= (t >> 1) & L;                           // -> A 0, C 1, G 2, T 3
    const uint64_t u = x | (0x20 * L);           // lower case
    auto eq = [&](unsigned char c) { const uint64_t v = u ^ (c * L); return ~(((v & M7) + M7) | v | M7); };   // 0x80 in the bytes equal to c (exact)
    const uint64_t inv = ~(eq('a') | eq('c') | eq('g') | eq('t') | eq('u')) & H;
    t &= ~((inv >> 7) * 3);
    t = (t | (t >> 6)) & 0x000F000F000F000FULL;
    t = (t | (t >> 12)) & 0x000000FF000000FFULL;
    t = (t | (t >> 24)) & 0xFFFFULL;
    code16 = (uint32_t)t;
    amb8 = (uint32_t)(((inv >> 7) * 0x0102040810204080ULL) >> 56);
}

// bit i of v -> bit 2i
static inline uint64_t spread32(uint64_t v)
{
    v = (v | v << 16) & 0x0000FFFF0000FFFFULL; v = (v | v << 8) & 0x00FF00FF00FF00FFULL; v = (v | v << 4) & 0x0F0F0F0F0F0F0F0FULL;
    v = (v | v << 2) & 0x3333333333333333ULL; v = (v | v << 1) & 0x5555555555555555ULL;
    return v;
}
// the same packing, 64 bases per call (4 code words + 2 ambiguity words), 32 per AVX2 step: the two code bits of a base
// are gathered with byte-mask moves and interleaved
__attribute__((target("avx2"))) static void pack_group64_avx2(const unsigned char *p, uint32_t *d2, uint32_t *dn)
{
    for (int h = 0; h < 2; ++h) {
        const __m256i x = _mm256_loadu_si256((const __m256i*)(p + 32 * h));
        const __m256i u = _mm256_or_si256(x, _mm256_set1_epi8(0x20));
        __m256i ok = _mm256_or_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8('a')), _mm256_cmpeq_epi8(u, _mm256_set1_epi8('c')));
        ok = _mm256_or_si256(ok, _mm256_cmpeq_epi8(u, _mm256_set1_epi8('g')));
        ok = _mm256_or_si256(ok, _mm256_cmpeq_epi8(u, _mm256_set1_epi8('t')));
        ok = _mm256_or_si256(ok, _mm256_cmpeq_epi8(u, _mm256_set1_epi8('u')));
        __m256i t = _mm256_and_si256(_mm256_srli_epi16(x, 1), _mm256_set1_epi8(3));                 // A 0, C 1, T 2, G 3
        t = _mm256_xor_si256(t, _mm256_and_si256(_mm256_srli_epi16(t, 1), _mm256_set1_epi8(1)));      // A 0, C 1, G 2, T 3
        t = _mm256_and_si256(t, ok);
        const uint64_t b0 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(t, 7)), b1 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(t, 6));
        const uint64_t code = spread32(b0) | spread32(b1) << 1;
        d2[2 * h] = (uint32_t)code; d2[2 * h + 1] = (uint32_t)(code >> 32);
        dn[h] = ~(uint32_t)_mm256_movemask_epi8(ok);
    }
}

// one sequence -> padded 64-base groups (host only, no device needed: the CPU test of the two packers)
static void pack_sequence(const unsigned char *p, int L, uint32_t *d2base, uint32_t *dnbase, bool avx2)
{
    for (int j0 = 0; j0 < L; j0 += 64) {           // one 64-base group = 4 code words + 2 ambiguity words
        uint32_t *d2 = d2base + (j0 >> 4), *dn = dnbase + (j0 >> 5);
        unsigned char tail[64];
        const unsigned char *g = p + j0;
        if (j0 + 64 > L) { memset(tail, 'A', 64); memcpy(tail, g, (size_t)(L - j0)); g = tail; }
        if (avx2) pack_group64_avx2(g, d2, dn);
        else {
            uint32_t cw[4] = { 0, 0, 0, 0 }, aw[2] = { 0, 0 };
            for (int k8 = 0; k8 < 8; ++k8) {
                uint64_t x8; memcpy(&x8, g + 8 * k8, 8);
                uint32_t code16, amb8;
                pack8(x8, code16, amb8);
                cw[k8 >> 1] |= code16 << ((k8 & 1) * 16);
                aw[k8 >> 2] |= amb8 << ((k8 & 3) * 8);
            }
            d2[0] = cw[0]; d2[1] = cw[1]; d2[2] = cw[2]; d2[3] = cw[3]; dn[0] = aw[0]; dn[1] = aw[1];
        }
    }
}
// debug tap: pack `len` bytes with the AVX2 (mode 1; TELR_E_ARG when the host lacks it) or the 64-bit word packer (mode 0)
// into code[(len+63)/64*4] and amb[(len+63)/64*2]
extern "C" int telr_debug_pack(const char *ascii, int32_t len, int mode, uint32_t *code, uint32_t *amb)
{
    if (!ascii || len < 0 || !code || !amb) return TELR_E_ARG;
    if (mode == 1 && !__builtin_cpu_supports("avx2")) return TELR_E_ARG;
    pack_sequence((const unsigned char*)ascii, len, code, amb, mode == 1);
    return TELR_OK;
}

extern "C" int telr_seqset_create(telr_ctx *ctx, int32_t n, const char *ascii, const int64_t *off, const int32_t *len, telr_seqset **out)
{
    (void)hipGetLastError();          // a failed allocation of an EARLIER call leaves its error with the thread: not this call's
    if (!ctx || n < 0 || !out || (n > 0 && (!ascii || !off || !len))) return TELR_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    telr_seqset *s = new telr_seqset();
    s->ctx = ctx; s->n = n; s->boff.resize(n + 1); s->len.assign(len, len + n);
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) {
        if (len[i] < 0) { delete s; return TELR_E_ARG; }
        s->boff[i] = tot; tot += ((int64_t)len[i] + 63) & ~63LL; s->total_bases += len[i];
        if (len[i] > s->max_len) s->max_len = len[i];
    }
    s->boff[n] = tot; s->padded_bases = tot;
    size_t w2 = (size_t)(tot / 16) + 8, wn = (size_t)(tot / 32) + 8;
    // Packing and upload are part of the host-inclusive rate of the path (a 30x read set is 4 GB of ASCII):
    //  * the two packed arrays are staged in PINNED grow-only buffers of the context (a fresh pageable buffer costs a page
    //    fault per 4 KB and a bounce copy inside the runtime), and every word of them is WRITTEN (whole 64-base groups,
    //    padding included), so they need no zero-fill;
    //  * the reads are packed in chunks of ~8 Mbases by worker threads (32 bases per AVX2 step where the host has it) while
    //    this thread sends the finished chunks, in order, with asynchronous copies: the transfer hides behind the packing.
    uint32_t *h2, *hn;
    { int r; if ((r = ctx_hbuf_t(ctx, "seq_stage2", w2, &h2)) != TELR_OK || (r = ctx_hbuf_t(ctx, "seq_stagen", wn, &hn)) != TELR_OK) { delete s; return r; } }
    for (size_t z = w2 - 8; z < w2; ++z) h2[z] = 0;
    for (size_t z = wn - 8; z < wn; ++z) hn[z] = 0;
    auto fail = [&](hipError_t e) { ctx->err = std::string("seqset upload: ") + hipGetErrorString(e); telr_seqset_free(s); return e == hipErrorOutOfMemory ? TELR_E_NOMEM : TELR_E_HIP; };
    hipError_t e;
    if ((e = hipMalloc(&s->d_seq2, w2 * 4)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&s->d_nmask, wn * 4)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&s->d_boff, (n + 1) * 8)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&s->d_len, (n ? n : 1) * 4)) != hipSuccess) return fail(e);
    std::vector<int32_t> cstart(1, 0);            // chunks of consecutive reads
    { int64_t acc = 0; for (int i = 0; i < n; ++i) { acc += ((int64_t)len[i] + 63) & ~63LL; if (acc >= (8LL << 20) && i + 1 < n) { cstart.push_back(i + 1); acc = 0; } } }
    cstart.push_back(n);
    const int nchunk = (int)cstart.size() - 1;
    std::unique_ptr<std::atomic<int>[]> done(new std::atomic<int>[nchunk > 0 ? nchunk : 1]);
    for (int c = 0; c < nchunk; ++c) done[c].store(0, std::memory_order_relaxed);
    std::atomic<int> next(0);
    static const bool avx2 = __builtin_cpu_supports("avx2") && !ab_on("no_avx2");
    int nth = (int)std::min<int64_t>(std::max(1u, std::thread::hardware_concurrency()), 16);
    if (const char *e = getenv("TELR_PACK_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) nth = v; }      // a caller that packs the next batch while the current one maps leaves cores to the mapper
    if (s->total_bases < (1 << 20)) nth = 1;
    if (nth > nchunk) nth = nchunk > 0 ? nchunk : 1;
    auto pack_chunk = [&](int c) {
        for (int i = cstart[c]; i < cstart[c + 1]; ++i)
            pack_sequence((const unsigned char*)ascii + off[i], len[i], h2 + (s->boff[i] >> 4), hn + (s->boff[i] >> 5), avx2);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nth && nchunk > 0; ++t) th.emplace_back([&]() {
        for (int c; (c = next.fetch_add(1)) < nchunk; ) { pack_chunk(c); done[c].store(1, std::memory_order_release); }
    });
    // send finished chunks in order, a few at a time (>= 16 MB of code words per copy)
    hipError_t ce = hipSuccess;
    { int c0 = 0;
      while (c0 < nchunk) {
          int c1 = c0; int64_t bases = 0;
          while (c1 < nchunk && (bases < (64LL << 20) || c1 == c0)) {
              while (!done[c1].load(std::memory_order_acquire)) std::this_thread::yield();
              bases += s->boff[cstart[c1 + 1]] - s->boff[cstart[c1]]; ++c1;
          }
          const int64_t b0 = s->boff[cstart[c0]], b1 = s->boff[cstart[c1]];
          if (ce == hipSuccess && b1 > b0) {
              ce = hipMemcpyAsync(s->d_seq2 + (b0 >> 4), h2 + (b0 >> 4), (size_t)((b1 - b0) >> 4) * 4, hipMemcpyHostToDevice, ctx->stream);
              if (ce == hipSuccess) ce = hipMemcpyAsync(s->d_nmask + (b0 >> 5), hn + (b0 >> 5), (size_t)((b1 - b0) >> 5) * 4, hipMemcpyHostToDevice, ctx->stream);
          }
          c0 = c1;
      } }
    for (auto &t : th) t.join();
    if (ce != hipSuccess) return fail(ce);
    if ((e = hipMemcpyAsync(s->d_seq2 + (tot >> 4), h2 + (tot >> 4), 8 * 4, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(s->d_nmask + (tot >> 5), hn + (tot >> 5), 8 * 4, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(s->d_boff, s->boff.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return fail(e);
    if (n && (e = hipMemcpyAsync(s->d_len, s->len.data(), n * 4, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return fail(e);
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(e);
    *out = s;
    return TELR_OK;
}
static int seqset_subset_impl(telr_ctx *ctx, const telr_seqset *parent, int32_t n, const int32_t *idx, const uint8_t *rc, telr_seqset **out)
{
    (void)hipGetLastError();          // a failed allocation of an EARLIER call leaves its error with the thread: not this call's
    if (!ctx || !parent || n < 0 || !out || (n > 0 && !idx)) return TELR_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    for (int i = 0; i < n; ++i) if (idx[i] < 0 || idx[i] >= parent->n) return TELR_E_ARG;
    telr_seqset *s = new telr_seqset();
    s->ctx = ctx; s->n = n; s->boff.resize(n + 1); s->len.resize(n);
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) {
        const int32_t L = parent->len[idx[i]];
        s->len[i] = L; s->boff[i] = tot; tot += ((int64_t)L + 63) & ~63LL; s->total_bases += L;
        if (L > s->max_len) s->max_len = L;
    }
    s->boff[n] = tot; s->padded_bases = tot;
    const size_t w2 = (size_t)(tot / 16) + 8, wn = (size_t)(tot / 32) + 8;
    auto fail = [&](hipError_t e) { ctx->err = std::string("seqset subset: ") + hipGetErrorString(e); telr_seqset_free(s); return e == hipErrorOutOfMemory ? TELR_E_NOMEM : TELR_E_HIP; };
    hipError_t e; int32_t *d_idx = nullptr; uint8_t *d_rc = nullptr;
    if ((e = hipMalloc(&s->d_seq2, w2 * 4)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&s->d_nmask, wn * 4)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&s->d_boff, (n + 1) * 8)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&s->d_len, (n ? n : 1) * 4)) != hipSuccess) return fail(e);
    // the index list lives in the context's scratch: a hipMalloc / hipFree pair per call would make the call wait for the whole device
    { void *p = nullptr; const int rc_ = ctx_buf(ctx, "subset_idx", (size_t)(n ? n : 1) * 4, &p); if (rc_ != TELR_OK) { telr_seqset_free(s); return rc_; } d_idx = (int32_t*)p; }
    if (rc) { void *p = nullptr; const int rc_ = ctx_buf(ctx, "subset_rc", (size_t)(n ? n : 1), &p); if (rc_ != TELR_OK) { telr_seqset_free(s); return rc_; } d_rc = (uint8_t*)p; }
    hipStream_t st = ctx->stream;
    // the 8 slack words behind the last sequence are read by window loads: keep them defined
    if ((e = hipMemsetAsync(s->d_seq2 + (w2 - 8), 0, 32, st)) != hipSuccess || (e = hipMemsetAsync(s->d_nmask + (wn - 8), 0, 32, st)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(s->d_boff, s->boff.data(), (n + 1) * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
    if (n) {
        if ((e = hipMemcpyAsync(s->d_len, s->len.data(), n * 4, hipMemcpyHostToDevice, st)) != hipSuccess || (e = hipMemcpyAsync(d_idx, idx, n * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
        if (d_rc && (e = hipMemcpyAsync(d_rc, rc, n, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
        hipLaunchKernelGGL(k_seq_gather, dim3(n), dim3(256), 0, st, parent->d_seq2, parent->d_nmask, parent->d_boff, d_idx, s->d_boff, n, s->d_seq2, s->d_nmask, (const uint8_t*)d_rc, (const int32_t*)s->d_len);
        if ((e = hipGetLastError()) != hipSuccess) return fail(e);
    }
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(e);
    *out = s;
    return TELR_OK;
}
extern "C" int telr_seqset_subset(telr_ctx *ctx, const telr_seqset *parent, int32_t n, const int32_t *idx, telr_seqset **out)
{
    return seqset_subset_impl(ctx, parent, n, idx, nullptr, out);
}
extern "C" int telr_seqset_subset_rc(telr_ctx *ctx, const telr_seqset *parent, int32_t n, const int32_t *idx, const uint8_t *rc, telr_seqset **out)
{
    return seqset_subset_impl(ctx, parent, n, idx, rc, out);
}
extern "C" int telr_seqset_packed(const telr_seqset *s, const void **d_seq2, const void **d_nmask, int64_t *nwords2, int64_t *nwordsn)
{
    if (!s || !d_seq2 || !d_nmask || !nwords2 || !nwordsn) return TELR_E_ARG;
    *d_seq2 = s->d_seq2; *d_nmask = s->d_nmask; *nwords2 = s->padded_bases / 16; *nwordsn = s->padded_bases / 32;
    return TELR_OK;
}
extern "C" int telr_seqset_from_packed(telr_ctx *ctx, int32_t n, const int32_t *len, const void *d_seq2, int64_t nwords2, const void *d_nmask, int64_t nwordsn, telr_seqset **out)
{
    (void)hipGetLastError();
    if (!ctx || n < 0 || !out || (n > 0 && !len)) return TELR_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    telr_seqset *s = new telr_seqset();
    s->ctx = ctx; s->n = n; s->boff.resize(n + 1); s->len.resize(n);
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) {
        const int32_t L = len[i];
        if (L < 0) { delete s; return TELR_E_ARG; }
        s->len[i] = L; s->boff[i] = tot; tot += ((int64_t)L + 63) & ~63LL; s->total_bases += L;
        if (L > s->max_len) s->max_len = L;
    }
    s->boff[n] = tot; s->padded_bases = tot;
    if (nwords2 != tot / 16 || nwordsn != tot / 32 || (tot > 0 && (!d_seq2 || !d_nmask))) { delete s; return TELR_E_ARG; }
    const size_t w2 = (size_t)(tot / 16) + 8, wn = (size_t)(tot / 32) + 8;
    auto fail = [&](hipError_t e) { ctx->err = std::string("seqset from packed words: ") + hipGetErrorString(e); telr_seqset_free(s); return e == hipErrorOutOfMemory ? TELR_E_NOMEM : TELR_E_HIP; };
    hipError_t e; hipStream_t st = ctx->stream;
    if ((e = hipMalloc(&s->d_seq2, w2 * 4)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&s->d_nmask, wn * 4)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&s->d_boff, (n + 1) * 8)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&s->d_len, (n ? n : 1) * 4)) != hipSuccess) return fail(e);
    if ((e = hipMemsetAsync(s->d_seq2 + (w2 - 8), 0, 32, st)) != hipSuccess || (e = hipMemsetAsync(s->d_nmask + (wn - 8), 0, 32, st)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(s->d_boff, s->boff.data(), (n + 1) * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
    if (n && (e = hipMemcpyAsync(s->d_len, s->len.data(), n * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e);
    if (tot) {
        if ((e = hipMemcpyAsync(s->d_seq2, d_seq2, (size_t)(tot / 16) * 4, hipMemcpyDeviceToDevice, st)) != hipSuccess) return fail(e);
        if ((e = hipMemcpyAsync(s->d_nmask, d_nmask, (size_t)(tot / 32) * 4, hipMemcpyDeviceToDevice, st)) != hipSuccess) return fail(e);
    }
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e);
    *out = s;
    return TELR_OK;
}
extern "C" void telr_seqset_free(telr_seqset *s)
{
    if (!s) return;
    (void)hipFree(s->d_seq2); (void)hipFree(s->d_nmask); (void)hipFree(s->d_boff); (void)hipFree(s->d_len);
    delete s;
}
extern "C" int64_t telr_seqset_bases(const telr_seqset *s) { return s ? s->total_bases : 0; }
extern "C" int32_t telr_seqset_count(const telr_seqset *s) { return s ? s->n : 0; }

// ---------------------------------------------------------------------------------------
// rocPRIM plumbing (scans and sorts are library calls; the hot kernels are in kernels.hip.h)
// out[0 .. n] = exclusive scan of cnt[0 .. n) (kernels.hip.h: k_qscan_sums / k_qscan_write -- hand-written, two small launches, no state to
// initialise); tot64 / n_over / over_list are optional
template <typename Tin, typename Tout>
static int dev_qscan(telr_ctx *ctx, const Tin *cnt, int32_t n, Tout *out, int64_t *tot64, int64_t cap, int32_t *n_over, int32_t *over_list)
{
    const int ntile = n > 0 ? (n + QSCAN_TILE - 1) / QSCAN_TILE : 1;
    int64_t *d_ts; TRY(ctx_buf_t(ctx, "qscan_tiles", (size_t)ntile, &d_ts));
    if (n_over) HIPCHK(hipMemsetAsync(n_over, 0, 4, ctx->stream));
    hipLaunchKernelGGL((k_qscan_sums<Tin>), dim3(ntile), dim3(256), 0, ctx->stream, cnt, n, cap, d_ts, n_over, over_list);
    hipLaunchKernelGGL((k_qscan_write<Tin, Tout>), dim3(ntile), dim3(256), 0, ctx->stream, cnt, n, d_ts, out, tot64);
    HIPCHK(hipGetLastError());
    return TELR_OK;
}
template <typename Tin, typename Tout>
static int dev_exclusive_scan(telr_ctx *ctx, const Tin *in, Tout *out, size_t n)
{
    if (n == 0) return TELR_OK;
    // round 6: every scan of the map path is the hand-written one (in[n - 1] is the callers' trailing zero: out[n - 1] = the total);
    // TELR_AB=scan_lib keeps rocPRIM's as the cross-check (tests/test_gpu_switches.py).  Up to 2^31 - 1 elements.
    static const bool scan_lib = ab_on("scan_lib");
    if (!scan_lib && n - 1 <= 0x7fffffff) return dev_qscan<Tin, Tout>(ctx, in, (int32_t)(n - 1), out, nullptr, 0, nullptr, nullptr);
    size_t tb = 0;
    auto it = rocprim::make_transform_iterator(in, [] __device__(Tin v) { return (Tout)v; });
    HIPCHK(rocprim::exclusive_scan(nullptr, tb, it, out, (Tout)0, n, rocprim::plus<Tout>(), ctx->stream));
    void *tmp; TRY(ctx_buf(ctx, "rp_tmp", tb, &tmp));
    HIPCHK(rocprim::exclusive_scan(tmp, tb, it, out, (Tout)0, n, rocprim::plus<Tout>(), ctx->stream));
    return TELR_OK;
}
static int dev_inclusive_scan_i32(telr_ctx *ctx, int32_t *io, size_t n)
{
    if (n == 0) return TELR_OK;
    size_t tb = 0;
    HIPCHK(rocprim::inclusive_scan(nullptr, tb, io, io, n, rocprim::plus<int32_t>(), ctx->stream));
    void *tmp; TRY(ctx_buf(ctx, "rp_tmp", tb, &tmp));
    HIPCHK(rocprim::inclusive_scan(tmp, tb, io, io, n, rocprim::plus<int32_t>(), ctx->stream));
    return TELR_OK;
}

// ---------------------------------------------------------------------------------------
// tiles for the sketch kernel
static TileList &ctx_tiles(telr_ctx *ctx) { return ctx->tiles; }
// tiles over slots; `slots_of(q)` = number of k-mer slots of sequence q; tile_seq holds q - seq_base
template <typename F> static void make_tiles_f(int32_t q0, int32_t q1, int32_t seq_base, F slots_of, TileList &T)
{
    T.seq.clear(); T.u0.clear(); T.first.assign(q1 - q0, 0);
    for (int32_t q = q0; q < q1; ++q) {
        T.first[q - q0] = (int32_t)T.seq.size();
        int ns = slots_of(q);
        for (int u = 0; u < ns; u += SK_TILE) { T.seq.push_back(q - seq_base); T.u0.push_back(u); }
    }
    T.n = (int32_t)T.seq.size();
}
static void make_tiles(const telr_seqset *s, int32_t q0, int32_t q1, int k, TileList &T)
{
    make_tiles_f(q0, q1, 0, [&](int32_t q) { return s->len[q] - k + 1; }, T);
}

// run the two-pass sketch over tiles; returns device arrays x,y (in ctx buffers named by prefix) and tile offsets
// staged = true: no compaction copy -- d_x / d_y come back as the per-tile staging arrays (tile t at t * SK_TILE) for a
// consumer that addresses them through the tile offsets (the seeding kernels)
static int run_sketch(telr_ctx *ctx, const telr_seqset *s, const TileList &T, int k, int w, const uint32_t *d_goff, const char *prefix,
                      uint64_t **d_x, uint32_t **d_y, int32_t **d_tile_off, int32_t *n_mz, const SketchHpcArgs *hpc = nullptr, bool staged = false)
{
    std::string P(prefix);
    int32_t *d_tseq, *d_tu0, *d_tcnt, *d_toff;
    TRY(ctx_buf_t(ctx, (P + "tile_seq").c_str(), T.n + 1, &d_tseq));
    TRY(ctx_buf_t(ctx, (P + "tile_u0").c_str(), T.n + 1, &d_tu0));
    TRY(ctx_buf_t(ctx, (P + "tile_cnt").c_str(), T.n + 1, &d_tcnt));
    TRY(ctx_buf_t(ctx, (P + "tile_off").c_str(), T.n + 1, &d_toff));
    *n_mz = 0; *d_tile_off = d_toff;
    if (T.n == 0) {
        TRY(ctx_buf_t(ctx, (P + "mz_x").c_str(), 1, d_x)); TRY(ctx_buf_t(ctx, (P + "mz_y").c_str(), 1, d_y));
        HIPCHK(hipMemsetAsync(d_toff, 0, 4, ctx->stream));
        return TELR_OK;
    }
    HIPCHK(hipMemcpyAsync(d_tseq, T.seq.data(), T.n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_tu0, T.u0.data(), T.n * 4, hipMemcpyHostToDevice, ctx->stream));
    SketchArgs A;
    A.seq2 = s->d_seq2; A.nmask = s->d_nmask; A.boff = s->d_boff; A.len = s->d_len; A.goff = d_goff;
    A.tile_seq = d_tseq; A.tile_u0 = d_tu0; A.k = k; A.w = w; A.tile_cnt = d_tcnt; A.tile_off = nullptr; A.out_x = nullptr; A.out_y = nullptr;
    SketchHpcArgs H;
    if (hpc) { H = *hpc; H.tile_seq = d_tseq; H.tile_u0 = d_tu0; H.k = k; H.w = w; H.tile_cnt = d_tcnt; H.tile_off = nullptr; H.out_x = nullptr; H.out_y = nullptr; }
    const int nslot = SK_TILE + 2 * (w - 1);
    size_t lds = hpc ? (size_t)nslot * 12 + (size_t)(nslot + k) + 32 + (size_t)((nslot + k) / 16 + 8) * 4 + (size_t)((nslot + k) / 32 + 8) * 4 : (size_t)nslot * 9 + 16;
    // single pass: every tile writes its minimizers into its own staging slots and its count; after the scan
    // of the counts a compaction copy packs them (one hash pass instead of a count pass + a write pass)
    uint64_t *d_sx; uint32_t *d_sy;
    TRY(ctx_buf_t(ctx, (P + "stg_x").c_str(), (size_t)T.n * SK_TILE, &d_sx));
    TRY(ctx_buf_t(ctx, (P + "stg_y").c_str(), (size_t)T.n * SK_TILE, &d_sy));
    if (hpc) { H.out_x = d_sx; H.out_y = d_sy; hipLaunchKernelGGL(k_sketch_hpc<2>, dim3(T.n), dim3(SK_THREADS), lds, ctx->stream, H); }
    else {
        A.out_x = d_sx; A.out_y = d_sy;
        if (k <= 15 && w == 10 && !ab_on("sketch64")) hipLaunchKernelGGL((k_sketch32<2, 9>), dim3(T.n), dim3(SK_THREADS), lds, ctx->stream, A);
        else if (k <= 15 && w == 5 && !ab_on("sketch64")) hipLaunchKernelGGL((k_sketch32<2, 4>), dim3(T.n), dim3(SK_THREADS), lds, ctx->stream, A);
        else if (k <= 15 && !ab_on("sketch64")) hipLaunchKernelGGL((k_sketch32<2, 0>), dim3(T.n), dim3(SK_THREADS), lds, ctx->stream, A);
        else hipLaunchKernelGGL(k_sketch<2>, dim3(T.n), dim3(SK_THREADS), lds, ctx->stream, A);
    }
    HIPCHK(hipGetLastError());
    // exclusive scan over T.n+1 entries so that tile_off[T.n] = total
    HIPCHK(hipMemsetAsync(d_tcnt + T.n, 0, 4, ctx->stream));
    TRY((dev_exclusive_scan<int32_t, int32_t>(ctx, d_tcnt, d_toff, (size_t)T.n + 1)));
    HIPCHK(hipMemcpyAsync(n_mz, d_toff + T.n, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (staged) { *d_x = d_sx; *d_y = d_sy; return TELR_OK; }
    TRY(ctx_buf_t(ctx, (P + "mz_x").c_str(), (size_t)*n_mz, d_x));
    TRY(ctx_buf_t(ctx, (P + "mz_y").c_str(), (size_t)*n_mz, d_y));
    hipLaunchKernelGGL(k_sketch_compact, dim3(T.n), dim3(256), 0, ctx->stream, d_sx, d_sy, d_tcnt, d_toff, *d_x, *d_y);
    HIPCHK(hipGetLastError());
    return TELR_OK;
}

// homopolymer compaction of sequences [q0,q1) of a set; fills the device part of `H` and the host run counts
static int build_hpc(telr_ctx *ctx, const telr_seqset *s, int32_t q0, int32_t q1, const char *prefix, SketchHpcArgs *H, std::vector<int32_t> *h_nrun)
{
    std::string P(prefix);
    const int nseq = q1 - q0;
    const int64_t chunk0 = s->boff[q0] / 64, chunk1 = s->boff[q1] / 64;
    const int nchunk = (int)(chunk1 - chunk0);
    uint64_t *d_flags; int32_t *d_cnt, *d_coff, *d_hoff; uint8_t *d_code; uint32_t *d_start;
    TRY(ctx_buf_t(ctx, (P + "hpc_flags").c_str(), (size_t)nchunk + 1, &d_flags));
    TRY(ctx_buf_t(ctx, (P + "hpc_cnt").c_str(), (size_t)nchunk + 1, &d_cnt));
    TRY(ctx_buf_t(ctx, (P + "hpc_coff").c_str(), (size_t)nchunk + 1, &d_coff));
    TRY(ctx_buf_t(ctx, (P + "hpc_hoff").c_str(), (size_t)nseq + 1, &d_hoff));
    int32_t total = 0;
    if (nchunk > 0) {
        hipLaunchKernelGGL(k_hpc_flags, dim3((nchunk + 255) / 256), dim3(256), 0, ctx->stream, s->d_seq2, s->d_nmask, s->d_boff, s->d_len, q0, nseq, chunk0, nchunk, d_flags, d_cnt);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemsetAsync(d_cnt + nchunk, 0, 4, ctx->stream));
        TRY((dev_exclusive_scan<int32_t, int32_t>(ctx, d_cnt, d_coff, (size_t)nchunk + 1)));
        HIPCHK(hipMemcpyAsync(&total, d_coff + nchunk, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    TRY(ctx_buf_t(ctx, (P + "hpc_code").c_str(), (size_t)total + 64, &d_code));
    TRY(ctx_buf_t(ctx, (P + "hpc_start").c_str(), (size_t)total + 64, &d_start));
    if (nchunk > 0) {
        hipLaunchKernelGGL(k_hpc_scatter, dim3((nchunk + 127) / 128), dim3(128), 0, ctx->stream, s->d_seq2, s->d_nmask, s->d_boff, q0, nseq, chunk0, nchunk, d_flags, d_coff, d_code, d_start);
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_hpc_seq_offsets, dim3((nseq + 256) / 256), dim3(256), 0, ctx->stream, s->d_boff, q0, nseq, chunk0, d_coff, nchunk, total, d_hoff);
    HIPCHK(hipGetLastError());
    std::vector<int32_t> hoff(nseq + 1);
    HIPCHK(hipMemcpyAsync(hoff.data(), d_hoff, (size_t)(nseq + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    h_nrun->resize(nseq);
    for (int i = 0; i < nseq; ++i) (*h_nrun)[i] = hoff[i + 1] - hoff[i];
    memset(H, 0, sizeof(*H));
    H->hcode = d_code; H->hstart = d_start; H->hoff = d_hoff; H->len = s->d_len + q0;
    return TELR_OK;
}

// ---------------------------------------------------------------------------------------
// index
struct telr_index {
    telr_ctx *ctx;
    const telr_seqset *targets;
    telr_idx_opt io;
    std::vector<uint32_t> goff;          // [n+1]
    int64_t n_mz = 0; int32_t n_ent = 0;
    int32_t bucket_bits = 0, shift = 0;
    uint64_t *d_ent_hash = nullptr; uint32_t *d_ent_off = nullptr, *d_pos = nullptr, *d_bstart = nullptr, *d_goff = nullptr;
    HtSlot *d_ht = nullptr; int32_t ht_shift = 0; uint32_t ht_mask = 0;       // probe table of the seeding kernel
    uint32_t *d_ht_home = nullptr;                                            // its home-slot bitmap
    std::vector<uint32_t> sorted_counts; // ascending, for the mid_occ quantile
    // per-target occurrence statistics (built on first use): runs = (minimizer, target) pairs, keys = target<<32 | count sorted
    mutable uint64_t *d_pt_keys = nullptr; mutable int32_t *d_pt_off = nullptr; mutable int32_t pt_runs = -1;
    mutable std::mutex pt_mu;                 // the lazily built per-target runs above: contexts may share an index (include/telr_hip.h)
    mutable double anchors_per_base = 0;  // seen by the last telr_map call on this index (0 = none yet): sizes the first range of the next call
    mutable double dens_bound = -1; mutable int32_t dens_mid = -1;     // index_density_bound() and the cut-off it was computed for
};

extern "C" void telr_index_free(telr_index *ix)
{
    if (!ix) return;
    (void)hipFree(ix->d_ent_hash); (void)hipFree(ix->d_ent_off); (void)hipFree(ix->d_pos); (void)hipFree(ix->d_bstart); (void)hipFree(ix->d_goff); (void)hipFree(ix->d_ht); (void)hipFree(ix->d_ht_home);
    (void)hipFree(ix->d_pt_keys); (void)hipFree(ix->d_pt_off);
    delete ix;
}
extern "C" int telr_index_stats(const telr_index *ix, int64_t *n_mz, int64_t *n_distinct)
{
    if (!ix) return TELR_E_ARG;
    if (n_mz) *n_mz = ix->n_mz;
    if (n_distinct) *n_distinct = ix->n_ent;
    return TELR_OK;
}

static int index_build_impl(telr_ctx *ctx, const telr_seqset *tg, const telr_idx_opt *io, telr_index *ix)
{
    const int n = tg->n, k = io->k, w = io->w;
    ix->ctx = ctx; ix->targets = tg; ix->io = *io;
    ix->goff.resize(n + 1);
    uint64_t g = 0;
    for (int i = 0; i < n; ++i) { ix->goff[i] = (uint32_t)g; g = (g + (uint64_t)tg->len[i] + TELR_TPAD + 63) & ~63ULL; if (g >= (1ULL << 31)) return TELR_E_RANGE; }
    ix->goff[n] = (uint32_t)g;
    HIPCHK(hipMalloc(&ix->d_goff, (n + 1) * 4));
    // (a null-stream copy on purpose: with the copy moved to the context's stream every later telr_map call on configs[2] ran 7 %
    // slower, 221 against 205 ms per step, A/B on one box with nothing else changed -- the process never touching the null stream
    // changes how the runtime schedules the context's blocking streams; the per-call copies of qtarget are stream-local)
    HIPCHK(hipMemcpy(ix->d_goff, ix->goff.data(), (n + 1) * 4, hipMemcpyHostToDevice));
    TileList T;
    uint64_t *d_x; uint32_t *d_y; int32_t *d_toff; int32_t nmz = 0;
    if (io->is_hpc) {
        SketchHpcArgs H; std::vector<int32_t> nrun;
        TRY(build_hpc(ctx, tg, 0, n, "ixh_", &H, &nrun));
        H.goff = ix->d_goff;
        make_tiles_f(0, n, 0, [&](int32_t q) { return nrun[q] - k + 1; }, T);
        TRY(run_sketch(ctx, tg, T, k, w, ix->d_goff, "ix_", &d_x, &d_y, &d_toff, &nmz, &H));
    } else {
        make_tiles(tg, 0, n, k, T);
        TRY(run_sketch(ctx, tg, T, k, w, ix->d_goff, "ix_", &d_x, &d_y, &d_toff, &nmz));
    }
    ix->n_mz = nmz;
    // hash = x >> 8 (in place), then stable radix sort by hash carrying y
    uint64_t *d_h2; uint32_t *d_y2;
    TRY(ctx_buf_t(ctx, "ix_h2", (size_t)nmz, &d_h2));
    HIPCHK(hipMalloc(&ix->d_pos, ((size_t)nmz + 1) * 4));
    d_y2 = ix->d_pos;
    // (the hand-written LSD radix sort of radix.hip.h; TELR_AB=index_sort_lib: rocPRIM's, the cross-check of tests/test_gpu_switches.py)
    static const bool sort_lib = ab_on("index_sort_lib");
    uint32_t *d_rshist = nullptr;
    if (nmz > 0 && !sort_lib) {
        const int passes = (2 * k + 7) / 8;
        uint64_t *d_tk; TRY(ctx_buf_t(ctx, "ix_rs_keys", (size_t)nmz, &d_tk));
        TRY(ctx_buf_t(ctx, "ix_rs_hist", (size_t)RS_BINS * ((nmz + RS_TILE - 1) / RS_TILE) + RS_BINS, &d_rshist));
        // an even number of passes ends where it began: the hashes start in the final arrays (d_h2, pos) then, else in the temporaries (d_tk, d_y)
        uint64_t *k0 = passes % 2 == 0 ? d_h2 : d_tk; uint32_t *v0 = passes % 2 == 0 ? d_y2 : d_y;
        hipLaunchKernelGGL(k_ix_hash_keys, dim3((nmz + 255) / 256), dim3(256), 0, ctx->stream, d_x, d_y, (int64_t)nmz, k0, v0);
        HIPCHK(hipGetLastError());
        if (passes % 2 == 0) (void)radix_sort_passes<uint64_t, true>(d_h2, d_y2, d_tk, d_y, (int64_t)nmz, 2 * k, d_rshist, ctx->stream);
        else (void)radix_sort_passes<uint64_t, true>(d_tk, d_y, d_h2, d_y2, (int64_t)nmz, 2 * k, d_rshist, ctx->stream);
        HIPCHK(hipGetLastError());
    } else if (nmz > 0) {
        auto hin = rocprim::make_transform_iterator(d_x, [] __device__(uint64_t v) { return v >> 8; });
        size_t tb = 0;
        HIPCHK(rocprim::radix_sort_pairs(nullptr, tb, hin, d_h2, d_y, d_y2, (size_t)nmz, 0, 2 * k, ctx->stream));
        void *tmp; TRY(ctx_buf(ctx, "rp_tmp", tb, &tmp));
        HIPCHK(rocprim::radix_sort_pairs(tmp, tb, hin, d_h2, d_y, d_y2, (size_t)nmz, 0, 2 * k, ctx->stream));
    }
    // distinct entries
    int32_t *d_flag, *d_rank;
    TRY(ctx_buf_t(ctx, "ix_flag", (size_t)nmz + 1, &d_flag));
    TRY(ctx_buf_t(ctx, "ix_rank", (size_t)nmz + 1, &d_rank));
    int32_t n_ent = 0;
    if (nmz > 0) {
        hipLaunchKernelGGL(k_head_flags, dim3((nmz + 255) / 256), dim3(256), 0, ctx->stream, d_h2, (int64_t)nmz, d_flag);
        HIPCHK(hipMemsetAsync(d_flag + nmz, 0, 4, ctx->stream));
        TRY((dev_exclusive_scan<int32_t, int32_t>(ctx, d_flag, d_rank, (size_t)nmz + 1)));
        HIPCHK(hipMemcpyAsync(&n_ent, d_rank + nmz, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    ix->n_ent = n_ent;
    HIPCHK(hipMalloc(&ix->d_ent_hash, ((size_t)n_ent + 1) * 8));
    HIPCHK(hipMalloc(&ix->d_ent_off, ((size_t)n_ent + 2) * 4));
    if (nmz > 0) {
        hipLaunchKernelGGL(k_write_entries, dim3((nmz + 255) / 256), dim3(256), 0, ctx->stream, d_h2, d_flag, d_rank, (int64_t)nmz, ix->d_ent_hash, ix->d_ent_off, n_ent);
        HIPCHK(hipGetLastError());
    } else HIPCHK(hipMemsetAsync(ix->d_ent_off, 0, 8, ctx->stream));
    // probe table for seeding: power-of-two slots, at least twice the distinct minimizers
    {
        int hb = 4; while ((1LL << hb) < 2LL * n_ent && hb < 30) ++hb;      // load <= 1/2, at least one empty slot
        ix->ht_shift = 64 - hb; ix->ht_mask = (uint32_t)((1ULL << hb) - 1);
        HIPCHK(hipMalloc(&ix->d_ht, ((size_t)1 << hb) * sizeof(HtSlot)));
        HIPCHK(hipMemsetAsync(ix->d_ht, 0xff, ((size_t)1 << hb) * sizeof(HtSlot), ctx->stream));
        const size_t home_words = (((size_t)1 << (hb + HT_FB_LOG)) + 31) / 32;
        HIPCHK(hipMalloc(&ix->d_ht_home, home_words * 4));
        HIPCHK(hipMemsetAsync(ix->d_ht_home, 0, home_words * 4, ctx->stream));
        if (n_ent > 0) hipLaunchKernelGGL(k_ht_build, dim3((n_ent + 255) / 256), dim3(256), 0, ctx->stream, ix->d_ent_hash, ix->d_ent_off, n_ent, ix->ht_shift, ix->ht_mask, ix->d_ht, ix->d_ht_home, ix->d_pos);
        HIPCHK(hipGetLastError());
    }
    // occurrence counts, sorted, to the host for the -f quantile
    ix->sorted_counts.resize(n_ent);
    if (n_ent > 0) {
        uint32_t *d_c, *d_c2;
        TRY(ctx_buf_t(ctx, "ix_cnt", (size_t)n_ent, &d_c));
        TRY(ctx_buf_t(ctx, "ix_cnt2", (size_t)n_ent, &d_c2));
        hipLaunchKernelGGL(k_ent_counts, dim3((n_ent + 255) / 256), dim3(256), 0, ctx->stream, ix->d_ent_off, n_ent, d_c);
        if (!sort_lib) {
            // a count is at most nmz: as many 8-bit passes as its bits need, an even number of them (the result then lies in d_c again)
            int nb = 8; while (nb < 32 && ((int64_t)1 << nb) <= (int64_t)nmz) nb += 8;
            if ((nb / 8) % 2) nb += 8;
            (void)radix_sort_passes<uint32_t, false>(d_c, nullptr, d_c2, nullptr, (int64_t)n_ent, nb, d_rshist, ctx->stream);
            HIPCHK(hipGetLastError());
            d_c2 = d_c;
        } else {
            size_t tb = 0;
            HIPCHK(rocprim::radix_sort_keys(nullptr, tb, d_c, d_c2, (size_t)n_ent, 0, 32, ctx->stream));
            void *tmp; TRY(ctx_buf(ctx, "rp_tmp", tb, &tmp));
            HIPCHK(rocprim::radix_sort_keys(tmp, tb, d_c, d_c2, (size_t)n_ent, 0, 32, ctx->stream));
        }
        HIPCHK(hipMemcpyAsync(ix->sorted_counts.data(), d_c2, (size_t)n_ent * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return TELR_OK;
}

extern "C" int telr_index_build(telr_ctx *ctx, const telr_seqset *targets, const telr_idx_opt *io, telr_index **out)
{
    (void)hipGetLastError();          // a failed allocation of an EARLIER call leaves its error with the thread: not this call's
    if (!ctx || !targets || !io || !out) return TELR_E_ARG;
    if (io->k < 4 || io->k > 28 || io->w < 1 || io->w > 255) return TELR_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    memset(ctx->stage_ms, 0, sizeof(ctx->stage_ms));
    telr_index *ix = new telr_index();
    StageTimer st(ctx, ST_INDEX, false);
    int r = index_build_impl(ctx, targets, io, ix);
    st.stop();
    if (r != TELR_OK) { telr_index_free(ix); return r; }
    *out = ix;
    return TELR_OK;
}

static int32_t index_mid_occ(const telr_index *ix, const telr_map_opt *mo)
{
    int64_t n = ix->n_ent; int32_t occ;
    if (n == 0) occ = mo->min_mid_occ;
    else {
        int64_t idx = (int64_t)((1.0 - (double)mo->mid_occ_frac) * (double)n);
        if (idx >= n) idx = n - 1;
        occ = (int32_t)ix->sorted_counts[idx] + 1;
    }
    if (occ < mo->min_mid_occ) occ = mo->min_mid_occ;
    if (mo->max_mid_occ > mo->min_mid_occ && occ > mo->max_mid_occ) occ = mo->max_mid_occ;
    return occ;
}

// Per-target occurrence cut-offs (device array [n_targets]) for `mo`: the statistics are built once per index, the
// cut-offs once per (f, U) setting.  Used when a query is confined to one target (qtarget) or chains are ranked per
// target (TELR_MF_PER_TARGET): both stand for the reference's one-aligner-run-per-contig call sites.
// Upper bound of the anchors a read base produces before any call has measured it: every read minimizer taken for an index
// minimizer (really one in two to five: sequencing errors), each bringing its occurrence count unless that exceeds the cut-off.
// 2 / (w + 1) minimizers per base x sum(c^2 | c <= cut-off) / sum(c).  It only has to keep the first ranges of the first call
// on an index within their anchor budget (13-mers on a 140-Mb genome: ~2 per base, where 15-mers give 0.3).
static double index_density_bound(const telr_index *ix, int32_t mid_occ)
{
    if (ix->dens_mid == mid_occ && ix->dens_bound >= 0) return ix->dens_bound;
    double s1 = 0, s2 = 0;
    for (uint32_t c : ix->sorted_counts) { s1 += c; if ((int64_t)c <= mid_occ) s2 += (double)c * c; }
    ix->dens_bound = s1 > 0 ? s2 / s1 * 2.0 / (ix->io.w + 1) : 0;
    ix->dens_mid = mid_occ;
    return ix->dens_bound;
}

static int index_per_target_occ(telr_ctx *ctx, const telr_index *ix, const telr_map_opt *mo, const int32_t **d_tmid)
{
    const int n = ix->targets->n; const int64_t nmz = ix->n_mz;
    hipStream_t st = ctx->stream;
    std::unique_lock<std::mutex> pt_lock(ix->pt_mu);
    if (ix->pt_runs < 0) {
        int32_t *d_head, *d_rank, *d_tid, *d_start; int32_t n_runs = 0;
        TRY(ctx_buf_t(ctx, "pt_head", (size_t)nmz + 1, &d_head));
        TRY(ctx_buf_t(ctx, "pt_rank", (size_t)nmz + 1, &d_rank));
        TRY(ctx_buf_t(ctx, "pt_tid", (size_t)nmz + 1, &d_tid));
        HIPCHK(hipMemsetAsync(d_head, 0, ((size_t)nmz + 1) * 4, st));
        if (nmz > 0) {
            hipLaunchKernelGGL(k_pt_entry_heads, dim3((ix->n_ent + 255) / 256), dim3(256), 0, st, ix->d_ent_off, ix->n_ent, d_head);
            hipLaunchKernelGGL(k_pt_run_heads, dim3((unsigned)((nmz + 255) / 256)), dim3(256), 0, st, ix->d_pos, nmz, ix->d_goff, n, d_head, d_tid);
            HIPCHK(hipGetLastError());
            TRY((dev_exclusive_scan<int32_t, int32_t>(ctx, d_head, d_rank, (size_t)nmz + 1)));
            HIPCHK(hipMemcpyAsync(&n_runs, d_rank + nmz, 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
        }
        TRY(ctx_buf_t(ctx, "pt_start", (size_t)n_runs + 1, &d_start));
        uint64_t *d_k0;
        TRY(ctx_buf_t(ctx, "pt_key0", (size_t)n_runs + 1, &d_k0));
        HIPCHK(hipMalloc(&ix->d_pt_keys, ((size_t)n_runs + 1) * 8));
        HIPCHK(hipMalloc(&ix->d_pt_off, ((size_t)n + 2) * 4));
        if (n_runs > 0) {
            hipLaunchKernelGGL(k_pt_run_starts, dim3((unsigned)((nmz + 255) / 256)), dim3(256), 0, st, d_head, d_rank, nmz, d_start, n_runs);
            hipLaunchKernelGGL(k_pt_run_keys, dim3((n_runs + 255) / 256), dim3(256), 0, st, d_start, d_tid, n_runs, d_k0);
            HIPCHK(hipGetLastError());
            size_t tb = 0;
            HIPCHK(rocprim::radix_sort_keys(nullptr, tb, d_k0, ix->d_pt_keys, (size_t)n_runs, 0, 64, st));
            void *tmp; TRY(ctx_buf(ctx, "rp_tmp", tb, &tmp));
            HIPCHK(rocprim::radix_sort_keys(tmp, tb, d_k0, ix->d_pt_keys, (size_t)n_runs, 0, 64, st));
        }
        hipLaunchKernelGGL(k_pt_target_off, dim3((n + 1 + 255) / 256), dim3(256), 0, st, ix->d_pt_keys, n_runs, n, ix->d_pt_off);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
        ix->pt_runs = n_runs;
    }
    pt_lock.unlock();
    // the cut-offs themselves depend on the call's options: the context's own buffer (two contexts may map against one index at once)
    int32_t *d_mid;
    TRY(ctx_buf_t(ctx, "pt_mid", (size_t)n + 1, &d_mid));
    hipLaunchKernelGGL(k_pt_mid_occ, dim3((n + 255) / 256), dim3(256), 0, st, ix->d_pt_keys, ix->d_pt_off, n, mo->mid_occ_frac, mo->min_mid_occ, mo->max_mid_occ, d_mid);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));       // worker contexts read it from their own streams
    *d_tmid = d_mid;
    return TELR_OK;
}

// ---------------------------------------------------------------------------------------
// host-side chain selection (the oracle's select_chains(), restated for the product)
struct Sel { int32_t ci, key, ord, fs, fe, tid, parent, subsc, n_sub, keep; };

// (same outcome as the oracle's quadratic loops; primaries and per-target tallies are kept in small side lists so that a
// read with thousands of repeat-induced chains stays linear in its chains times its primaries)
static void select_chains(std::vector<Sel> &s, const telr_map_opt *mo, const std::vector<int32_t> &sub_score)
{
    const bool per_t = (mo->flags & TELR_MF_PER_TARGET) != 0;
    const int n = (int)s.size();
    std::vector<int32_t> prim;                       // indices of the primaries found so far, in rank order
    for (int i = 0; i < n; ++i) {
        s[i].parent = i; s[i].subsc = 0; s[i].n_sub = 0;
        for (int j : prim) {
            if (per_t && s[j].tid != s[i].tid) continue;
            int32_t lo = std::max(s[i].fs, s[j].fs), hi = std::min(s[i].fe, s[j].fe);
            int32_t ol = hi > lo ? hi - lo : 0;
            int32_t mn = std::min(s[i].fe - s[i].fs, s[j].fe - s[j].fs);
            if ((float)ol > mo->mask_level * (float)mn) {
                s[i].parent = j;
                if (sub_score[s[i].ci] > s[j].subsc) s[j].subsc = sub_score[s[i].ci];
                ++s[j].n_sub;
                break;
            }
        }
        if (s[i].parent == i) prim.push_back(i);
    }
    int n2_all = 0; std::vector<std::pair<int32_t, int32_t>> n2_tid;     // kept secondaries so far (per target with PER_TARGET)
    for (int i = 0; i < n; ++i) {
        if (s[i].parent == i) { s[i].keep = 1; continue; }
        s[i].keep = 0;
        if (!mo->secondary) continue;
        if ((float)s[i].key < (float)s[s[i].parent].key * mo->pri_ratio) continue;
        int *cnt = &n2_all;
        if (per_t) {
            size_t z = 0;
            while (z < n2_tid.size() && n2_tid[z].first != s[i].tid) ++z;
            if (z == n2_tid.size()) n2_tid.push_back(std::make_pair(s[i].tid, 0));
            cnt = &n2_tid[z].second;
        }
        if (*cnt < mo->best_n) { s[i].keep = 1; ++*cnt; }
    }
}
static bool sel_less(const Sel &x, const Sel &y) { return x.key != y.key ? x.key > y.key : x.ord < y.ord; }

static int32_t mapq_of(const telr_aln &r, const telr_map_opt *mo)
{
    if (!(r.flags & TELR_F_PRIMARY) && !(r.flags & TELR_F_SUPPL)) return 0;
    float f1 = (float)r.score, f2 = (float)(r.subsc > mo->min_chain_score ? r.subsc : mo->min_chain_score);
    float pen_cm = r.cnt > 10 ? 1.0f : 0.1f * (float)r.cnt;
    float x = f2 / f1; if (x > 1.0f) x = 1.0f;
    int32_t mq = (int32_t)(40.0f * (1.0f - x) * pen_cm * logf(f1));
    if (mq > 60) mq = 60;
    if (mq < 0) mq = 0;
    return mq;
}

struct telr_result {
    telr_ctx *ctx = nullptr;
    std::vector<telr_aln> alns;
    uint32_t *cig = nullptr;       // pinned; filled by one DMA that may still be in flight when telr_map returns
    size_t ncig = 0, cap = 0;      // cap in ops
    mutable hipEvent_t dma_done = nullptr;   // non-null while the CIGAR DMA has not been waited for
    // TELR_MF_KEEP_CIGARS: the same array on the device (same offsets), for telr_write_bam_dev.  Mirrored piece by piece as the
    // host array grows: the stitched scratch of a range by a device copy.  Complete iff twin_n == ncig and !twin_off.
    uint32_t *d_cig = nullptr; size_t d_cap = 0, twin_n = 0; bool twin_off = false;
    mutable std::vector<hipEvent_t> up_events;   // uploads into d_cig that read `cig`: waited for before `cig` moves
    // Ranges of one telr_map call may be in flight on two slots (telr_map: range pipelining).  They append to this result in
    // range order: a range waits here for its turn before it first touches alns / cig, and keeps the turn until it is in.
    // turn < 0: a range failed, everybody leaves.
    std::mutex gate_m; std::condition_variable gate_cv; int turn = 0;
    ~telr_result();
};
// what a batch saw when it got its turn
struct RangeTurn { int turn = -1; bool entered = false; size_t a0 = 0, c0 = 0; };
static bool gate_enter(telr_result *R, RangeTurn *g)
{
    if (!g || g->entered) return true;
    if (g->turn >= 0) {
        std::unique_lock<std::mutex> lk(R->gate_m);
        R->gate_cv.wait(lk, [&] { return R->turn < 0 || R->turn == g->turn; });
        if (R->turn < 0) return false;
    }
    g->entered = true; g->a0 = R->alns.size(); g->c0 = R->ncig;
    return true;
}
static void gate_leave(telr_result *R, int turn, bool ok)
{
    if (turn < 0) return;
    { std::lock_guard<std::mutex> lk(R->gate_m); if (R->turn >= 0) R->turn = ok ? turn + 1 : -1; }
    R->gate_cv.notify_all();
}
// the records are complete when telr_map returns; the CIGAR array is complete after this (every accessor of it calls it)
static void result_wait(const telr_result *r)
{
    if (r && r->dma_done) { (void)hipEventSynchronize(r->dma_done); (void)hipEventDestroy(r->dma_done); r->dma_done = nullptr; }
    if (r) { for (hipEvent_t e : r->up_events) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); } r->up_events.clear(); }
}
extern "C" int telr_result_wait(const telr_result *r) { if (!r) return TELR_E_ARG; result_wait(r); return TELR_OK; }
// CIGAR buffers of freed results are kept (at most two) and handed to the next telr_map call: a fresh
// 200 MB allocation costs ~20 ms of page faults and another ~20 ms of munmap per call.
// result CIGAR buffers are PINNED host memory (the stitched CIGARs are copied device -> result in one DMA)
static uint32_t *cig_alloc(size_t ops) { void *p = nullptr; return hipHostMalloc(&p, ops * 4, hipHostMallocDefault) == hipSuccess ? (uint32_t*)p : nullptr; }
static void cig_free(uint32_t *p) { if (p) (void)hipHostFree(p); }
// grow to at least `want` ops keeping the first `keep` ops
static bool cig_grow(uint32_t **p, size_t *cap, size_t keep, size_t want)
{
    if (want <= *cap) return true;
    uint32_t *n = cig_alloc(want);
    if (!n) return false;
    if (*p && keep) memcpy(n, *p, keep * 4);
    cig_free(*p);
    *p = n; *cap = want;
    return true;
}
static void pool_put(telr_ctx *ctx, uint32_t *p, size_t cap)
{
    if (!p) return;
    if (!ctx || ctx->cig_pool.size() >= 2) { cig_free(p); return; }
    ctx->cig_pool.push_back(std::make_pair(p, cap));
}
// mirror words [base, base + n) of a result's CIGAR array on the device; src_dev: device source (the stitched scratch, copied on
// `st`), else the host array itself is uploaded on `st` and the event kept (the host array must not move under the upload).
// owner: the top-level context of the call (its pool supplies / takes the device array); bases: query bases of the whole call.
static void twin_put(telr_ctx *owner, telr_result *R, const telr_map_opt *mo, size_t base, size_t n, const uint32_t *src_dev, int64_t bases, hipStream_t st)
{
    if (!(mo->flags & TELR_MF_KEEP_CIGARS) || R->twin_off) return;
    if (base != R->twin_n) { R->twin_off = true; return; }           // a piece went by unmirrored
    if (!R->d_cig) {
        const size_t want = std::max<size_t>((size_t)((double)bases * 0.3) + ((size_t)1 << 20), base + n + 1);
        if (owner && owner->twin_pool && owner->twin_pool_cap >= want) { R->d_cig = owner->twin_pool; R->d_cap = owner->twin_pool_cap; owner->twin_pool = nullptr; owner->twin_pool_cap = 0; }
        else {
            if (owner && owner->twin_pool) { (void)hipFree(owner->twin_pool); owner->twin_pool = nullptr; owner->twin_pool_cap = 0; }
            if (hipMalloc(&R->d_cig, want * 4) != hipSuccess) { (void)hipGetLastError(); R->d_cig = nullptr; R->twin_off = true; return; }
            R->d_cap = want;
        }
    }
    if (base + n > R->d_cap) { R->twin_off = true; return; }         // more ops per base than any preset has produced: the writer uploads instead
    if (n) {
        if (src_dev) { if (hipMemcpyAsync(R->d_cig + base, src_dev, n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) { R->twin_off = true; return; } }
        else {
            hipEvent_t e;
            if (hipMemcpyAsync(R->d_cig + base, R->cig + base, n * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { R->twin_off = true; return; }
            (void)hipEventRecord(e, st); R->up_events.push_back(e);
        }
    }
    R->twin_n = base + n;
}
static void pool_get(telr_ctx *ctx, uint32_t **p, size_t *cap)
{
    *p = nullptr; *cap = 0;
    if (!ctx || ctx->cig_pool.empty()) return;
    size_t best = 0;
    for (size_t i = 1; i < ctx->cig_pool.size(); ++i) if (ctx->cig_pool[i].second > ctx->cig_pool[best].second) best = i;
    *p = ctx->cig_pool[best].first; *cap = ctx->cig_pool[best].second;
    ctx->cig_pool.erase(ctx->cig_pool.begin() + best);
}
telr_result::~telr_result()
{
    result_wait(this); pool_put(ctx, cig, cap);
    if (d_cig) { if (ctx && !ctx->twin_pool) { ctx->twin_pool = d_cig; ctx->twin_pool_cap = d_cap; } else (void)hipFree(d_cig); d_cig = nullptr; }
    // (a worker context never takes from its pool -- its results are created without one -- so it does not keep vectors either)
    if (ctx && alns.capacity() && ctx->aln_pool.size() < 2) { alns.clear(); ctx->aln_pool.emplace_back(std::move(alns)); }
}
// a result made of caller-supplied records and CIGAR words (copied): one rank writes the BAM of reads other ranks mapped
extern "C" int telr_result_from_arrays(telr_ctx *ctx, const telr_aln *alns, int64_t n, const uint32_t *cigars, int64_t n_cigar, telr_result **out)
{
    if (!ctx || !out || n < 0 || n_cigar < 0 || (n > 0 && !alns) || (n_cigar > 0 && !cigars)) return TELR_E_ARG;
    for (int64_t i = 0; i < n; ++i) if (alns[i].n_cigar < 0 || alns[i].cigar_off < 0 || alns[i].cigar_off + alns[i].n_cigar > n_cigar) return TELR_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    telr_result *R = new telr_result();
    R->ctx = ctx;
    R->alns.assign(alns, alns + n);
    R->cig = cig_alloc((size_t)n_cigar + 1);
    if (!R->cig) { delete R; return TELR_E_NOMEM; }
    R->cap = (size_t)n_cigar + 1; R->ncig = (size_t)n_cigar;
    if (n_cigar) memcpy(R->cig, cigars, (size_t)n_cigar * 4);
    *out = R;
    return TELR_OK;
}
// the same with the CIGAR words on the DEVICE (what an all-to-all delivered): they become the result's device copy as they are
// (the device BAM writer reads them in place) and are mirrored to the host array once
extern "C" int telr_result_from_device_cigars(telr_ctx *ctx, const telr_aln *alns, int64_t n, const void *d_cigars, int64_t n_cigar, telr_result **out)
{
    if (!ctx || !out || n < 0 || n_cigar < 0 || (n > 0 && !alns) || (n_cigar > 0 && !d_cigars)) return TELR_E_ARG;
    for (int64_t i = 0; i < n; ++i) if (alns[i].n_cigar < 0 || alns[i].cigar_off < 0 || alns[i].cigar_off + alns[i].n_cigar > n_cigar) return TELR_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    telr_result *R = new telr_result();
    R->ctx = ctx;
    R->alns.assign(alns, alns + n);
    R->cig = cig_alloc((size_t)n_cigar + 1);
    if (!R->cig) { delete R; return TELR_E_NOMEM; }
    R->cap = (size_t)n_cigar + 1; R->ncig = (size_t)n_cigar;
    if (hipMalloc(&R->d_cig, ((size_t)n_cigar + 1) * 4) != hipSuccess) { (void)hipGetLastError(); delete R; return TELR_E_NOMEM; }
    R->d_cap = (size_t)n_cigar + 1;
    if (n_cigar) {
        hipError_t e = hipMemcpy(R->d_cig, d_cigars, (size_t)n_cigar * 4, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(R->cig, R->d_cig, (size_t)n_cigar * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { ctx->err = hipGetErrorString(e); delete R; return TELR_E_HIP; }
    }
    R->twin_n = R->ncig; R->twin_off = false;
    *out = R;
    return TELR_OK;
}
extern "C" int64_t telr_result_count(const telr_result *r) { return r ? (int64_t)r->alns.size() : 0; }
extern "C" const telr_aln *telr_result_alns(const telr_result *r) { return r ? r->alns.data() : nullptr; }
// test tap: the device copy of the CIGAR array kept under TELR_MF_KEEP_CIGARS -> `out` (n words); returns the number of words
// copied, -1 when the result has no complete device copy
extern "C" int64_t telr_debug_result_twin(const telr_result *r, uint32_t *out, int64_t n)
{
    if (!r || !out) return -1;
    result_wait(r);
    if (!r->d_cig || r->twin_off || r->twin_n != r->ncig || (int64_t)r->ncig > n) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (r->ncig && hipMemcpy(out, r->d_cig, r->ncig * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)r->ncig;
}
extern "C" int64_t telr_result_cigar_count(const telr_result *r) { return r ? (int64_t)r->ncig : 0; }
extern "C" const uint32_t *telr_result_cigars(const telr_result *r) { result_wait(r); return r ? r->cig : nullptr; }
extern "C" void telr_result_free(telr_result *r) { delete r; }


// worker threads for the host phases: 1.5x the CPUs this process may actually use (cgroup v2 quota when
// present: the MI355X box reports 256 hardware threads but runs under cpu.max = 16 CPUs), at most 48
static int host_threads()
{
    static int cached = 0;
    if (cached) return cached;
    int n = (int)std::thread::hardware_concurrency();
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64]; long period = 0;
        if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            long quota = atol(q);
            int c = (int)((quota + period - 1) / period);
            if (c >= 1 && c < n) n = c;
        }
        fclose(f);
    }
    n = n + n / 2;
    // one process per GPU (torch.distributed.run sets LOCAL_WORLD_SIZE): the ranks of a node share its CPUs
    if (const char *e = getenv("LOCAL_WORLD_SIZE")) { int v = atoi(e); if (v > 1) n = std::max(4, n / v); }
    if (const char *e = getenv("TELR_HOST_THREADS")) { int v = atoi(e); if (v > 0) n = v; }
    if (n < 1) n = 1;
    if (n > 48) n = 48;
    cached = n;
    return n;
}
// munmap of a large populated mapping keeps the address-space lock for its whole page-table walk (~50 ms per GB): every other
// mmap / munmap / hipFree / fork of the process waits behind it.  Piece by piece, the others get their turn in between.
static void unmap_in_pieces(void *p, size_t bytes)
{
    const size_t piece = (size_t)32 << 20;
    for (size_t o = 0; o < bytes; o += piece) { munmap((char*)p + o, std::min(piece, bytes - o)); std::this_thread::yield(); }
}
// Persistent worker pool for the host phases (spawning ~24 threads per phase costs more than some of the phases).
// One job at a time; run() hands out task indices 0..n-1 to the workers and the caller and returns when all are done.
#include <condition_variable>
#include <functional>
#include <mutex>
#include <atomic>
class HostPool {
public:
    static HostPool &get() { static HostPool p; return p; }
    void run(int ntask, const std::function<void(int)> &f)
    {
        if (ntask <= 0) return;
        std::unique_lock<std::mutex> job(job_mu_);          // serialise callers (the two range slots share the pool)
        ensure(ntask - 1);
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &f; ntask_ = ntask; next_.store(0); pending_ = ntask; ++gen_;
        }
        cv_.notify_all();
        work();
        // every task done AND every worker back outside work(): nobody can touch the next job's counters with stale state
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0 && active_ == 0; });
        fn_ = nullptr;
    }
private:
    HostPool() {}
    ~HostPool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void ensure(int n) { while ((int)th_.size() < n && (int)th_.size() < 47) th_.emplace_back([this, g = gen_]() mutable { loop(g); }); }
    void work()
    {
        for (;;) {
            const int i = next_.fetch_add(1);
            if (i >= ntask_) break;
            (*fn_)(i);
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void loop(uint64_t seen)
    {
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                if (pending_ == 0) continue;        // woke up after the job was already finished by others
                ++active_;
            }
            work();
            std::lock_guard<std::mutex> lk(mu_);
            if (--active_ == 0 && pending_ == 0) done_.notify_all();
        }
    }
    std::mutex mu_, job_mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> th_;
    const std::function<void(int)> *fn_ = nullptr;
    std::atomic<int> next_{0};
    int ntask_ = 0, pending_ = 0, active_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};
// run f(t, begin, end) over [0,n) split into nt contiguous ranges (in order of t)
template <typename F> static void parallel_ranges(int nt, int n, F f)
{
    if (n <= 0) return;
    if (nt > n) nt = n;
    if (nt <= 1 || n < 64) { f(0, 0, n); return; }
    HostPool::get().run(nt, [&](int t) {
        const int a = (int)((int64_t)n * t / nt), b = (int)((int64_t)n * (t + 1) / nt);
        f(t, a, b);
    });
}


// ---------------------------------------------------------------------------------------
// One DP pass over a problem array: scratch sizing, class lists (sorted by length so that the problems
// sharing a wave are alike and the long ones start first), forward kernels, trace-back.
// longest gap fill (m+n) the packed int16 kernels take: |H| <= b*(m+n)/2 + q2 + 128*e2 and a*(m+n)/2 must stay inside
// +-16000 (bands of these classes have at most 128 diagonals); 0 disables the packed classes
// convex gap cost (cx_scale > 0): the packed int16 classes re-bias their scores as they go (kernels.hip.h, REB), so the length of
// a fill does not matter; what must hold is (1) the packed constants fit, (2) a live cell cannot sink out of int16 between two
// rounds (2 (open + ext_max) a trip, 32 trips), (3) the 12,000-wide window holds every cell an optimal path can use in a band
// of D diagonals: one gap across the band + the matches skipped meanwhile, gap(D) + a S D / 2 (derivation: kernels.hip.h, REB).
static inline int cx_gap(const telr_map_opt *mo, int L)
{
    int t = mo->cx_open;
    for (int i = 0; i < L; ++i) { const int e = mo->cx_ext_max - mo->cx_decay * i; t += e > mo->cx_ext_min ? e : mo->cx_ext_min; }
    return t;
}
static inline bool pk_cx_ok(const telr_map_opt *mo, int D)
{
    const int S = mo->cx_scale, bS = mo->b * S, aS = mo->a * S;
    if (ab_on("no_pk") || bS > 400 || aS > 400 || mo->sc_ambi * S > 400 || mo->cx_open + mo->cx_ext_max > 320 || mo->cx_ext_max < mo->cx_ext_min || mo->cx_ext_min < 0 || mo->cx_decay < 0) return false;
    return cx_gap(mo, D) + aS * D / 2 <= 11900;
}
static inline int pk_steps_limit(const telr_map_opt *mo)
{
    // (convex cost: the single-wave packed classes re-bias their scores as they go -- kernels.hip.h, REB -- so only the constants have to fit)
    if (mo->cx_scale > 0) return pk_cx_ok(mo, 128) ? 7900 : 0;
    if (!(mo->b <= 9 && mo->a <= 4 && mo->q2 + mo->e2 <= 64 && mo->sc_ambi <= 9) || ab_on("no_pk")) return 0;
    const int by_b = 2 * (15800 - mo->q2 - 128 * mo->e2) / (mo->b > 0 ? mo->b : 1) - 2, by_a = 32000 / (mo->a > 0 ? mo->a : 1) - 2;
    const int lim = by_b < by_a ? by_b : by_a;
    return lim > 0 ? lim : 0;
}
// same bound for the wide int16 classes (bands up to 1024 diagonals)
static inline int pk_wide_limit(const telr_map_opt *mo)
{
    if (!pk_steps_limit(mo) || ab_on("no_pkw")) return 0;
    if (mo->cx_scale > 0) return pk_cx_ok(mo, 256) ? 7900 : 0;          // (which of the three wide classes: pk_wide_maxd)
    const int by_b = 2 * (15800 - mo->q2 - 1024 * mo->e2) / (mo->b > 0 ? mo->b : 1) - 2, by_a = 32000 / (mo->a > 0 ? mo->a : 1) - 2;
    const int lim = by_b < by_a ? by_b : by_a;
    return lim > 0 ? lim : 0;
}
// the widest band the wide int16 classes take (19: 256, 20: 512, 21: 1,024 diagonals)
static inline int pk_wide_maxd(const telr_map_opt *mo)
{
    if (mo->cx_scale <= 0) return 1024;
    return pk_cx_ok(mo, 1024) ? 1024 : pk_cx_ok(mo, 512) ? 512 : pk_cx_ok(mo, 256) ? 256 : 0;
}
// longest z-drop extension window (m+n) the packed int16 kernel takes: scores stay inside +-16000
// classes that spill four bits per cell (kernels.hip.h: d_tb4): the one-piece classes of the preset, when every class has its
// own trace-back launch (the one-launch walk of TELR_AB=tb_one_launch reads bytes); TELR_AB=tb8 keeps the byte spill for A/B
static inline int tb4_mask(const telr_map_opt *mo)
{
    static const bool off = ab_on("tb_one_launch") || ab_on("tb8");
    if (off || !pk_steps_limit(mo) || mo->cx_scale > 0) return 0;          // (the convex cell spills plain bytes)
    const int d = d_onep_d(mo->q, mo->e, mo->q2, mo->e2);
    return (d >= 16 ? 1 : 0) | (d >= 20 ? 2 : 0);
}
// steps (m + n) up to which the nibble cell's quarter of the int16 range holds: 4 (b steps / 2 + q + e D) stays below 15,400
static inline int tb4_steps(const telr_map_opt *mo)
{
    const int by_b = 2 * (3850 - mo->q - 32 * mo->e) / (mo->b > 0 ? mo->b : 1) - 2, by_a = 7700 / (mo->a > 0 ? mo->a : 1) - 2;
    const int lim = by_b < by_a ? by_b : by_a;
    return lim > 0 ? lim : 0;
}
// the same for the two-piece tagged cell (scores times eight); off with the one-launch trace-back (it reads plain flags) or TELR_AB=no_tag8
static inline int tag8_steps(const telr_map_opt *mo)
{
    static const bool off = ab_on("tb_one_launch") || ab_on("no_tag8");
    if (off || !pk_steps_limit(mo) || mo->cx_scale > 0) return 0;
    const int by_b = 2 * (1975 - mo->q2 - 128 * mo->e2) / (mo->b > 0 ? mo->b : 1) - 2, by_a = 4000 / (mo->a > 0 ? mo->a : 1) - 2;
    const int lim = by_b < by_a ? by_b : by_a;
    return lim > 0 ? lim : 0;
}
// diagonals of the preset's extension band (-ext_band rounded to even .. ext_band), as the packed extension classes cut them: 64 / 128 / 256
static inline int pk_ext_d(const telr_map_opt *mo) { const int D = 2 * mo->ext_band + 2; return D <= 64 ? 64 : D <= 128 ? 128 : 256; }
static inline int pk_ext_limit(const telr_map_opt *mo)
{
    if (!pk_steps_limit(mo) || ab_on("no_pkext")) return 0;
    const int D = pk_ext_d(mo);
    // (convex cost: the z-drop test runs on the re-biased row maximum + the sum of the moves, in int32 -- kernels.hip.h, REB)
    if (mo->cx_scale > 0) return mo->zdrop * mo->cx_scale <= 30000 && pk_cx_ok(mo, D) ? 7900 : 0;
    if (mo->zdrop > 4000) return 0;
    const int by_b = 2 * (15800 - mo->q2 - D * mo->e2) / (mo->b > 0 ? mo->b : 1) - 2, by_a = 32000 / (mo->a > 0 ? mo->a : 1) - 2;
    const int lim = by_b < by_a ? by_b : by_a;
    return lim > 0 ? lim : 0;
}
// the widest band the packed extension classes take (18: 64, 23: 128, 24: 256 diagonals)
static inline int pk_ext_maxd(const telr_map_opt *mo) { return pk_ext_limit(mo) ? pk_ext_d(mo) : 64; }
static int dp_pass(telr_ctx *ctx, const telr_seqset *qs, const telr_seqset *tg, const telr_map_opt *mo, DpProb *d_probs, int np, DpRes *d_res,
                   uint32_t **d_rawcig_io, int32_t *d_retry, const std::string &sfx, bool primary)
{
    hipStream_t st = ctx->stream;
    int64_t *d_tbb, *d_cgo, *d_tboff, *d_cgoff; int32_t *d_clscnt, *d_clslist; uint32_t *d_clskey, *d_keytmp; int32_t *d_listtmp;
    ClsOff coff;
    TRY(ctx_buf_t(ctx, ("tb_bytes" + sfx).c_str(), (size_t)np + 1, &d_tbb));
    TRY(ctx_buf_t(ctx, ("cig_ops" + sfx).c_str(), (size_t)np + 1, &d_cgo));
    TRY(ctx_buf_t(ctx, ("tb_off" + sfx).c_str(), (size_t)np + 1, &d_tboff));
    TRY(ctx_buf_t(ctx, ("cig_off" + sfx).c_str(), (size_t)np + 1, &d_cgoff));
    TRY(ctx_buf_t(ctx, ("cls_cnt" + sfx).c_str(), 32, &d_clscnt));
    TRY(ctx_buf_t(ctx, ("cls_list" + sfx).c_str(), (size_t)np, &d_clslist));
    TRY(ctx_buf_t(ctx, ("cls_key" + sfx).c_str(), (size_t)np, &d_clskey));
    TRY(ctx_buf_t(ctx, ("cls_keytmp" + sfx).c_str(), (size_t)np, &d_keytmp));
    TRY(ctx_buf_t(ctx, ("cls_listtmp" + sfx).c_str(), (size_t)np, &d_listtmp));
    hipLaunchKernelGGL(k_prob_sizes, dim3((np + 255) / 256), dim3(256), 0, st, d_probs, np, primary ? mo->fill_margin : 0, pk_steps_limit(mo), pk_ext_limit(mo), pk_wide_limit(mo), qs->d_nmask, tg->d_nmask, d_tbb, d_cgo, tb4_mask(mo), tb4_steps(mo), pk_wide_maxd(mo), pk_ext_maxd(mo));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(d_tbb + np, 0, 8, st));
    HIPCHK(hipMemsetAsync(d_cgo + np, 0, 8, st));
    TRY((dev_exclusive_scan<int64_t, int64_t>(ctx, d_tbb, d_tboff, (size_t)np + 1)));
    if (primary) TRY((dev_exclusive_scan<int64_t, int64_t>(ctx, d_cgo, d_cgoff, (size_t)np + 1)));
    HIPCHK(hipMemsetAsync(d_clscnt, 0, 128, st));
    hipLaunchKernelGGL(k_prob_assign, dim3((np + 255) / 256), dim3(256), 0, st, d_probs, np, d_tboff, primary ? d_cgoff : (const int64_t*)nullptr, d_clscnt, d_clskey, d_listtmp);
    HIPCHK(hipGetLastError());
    int64_t tb_total = 0, cg_total = 0; int32_t h_cls[32];
    static_assert(DP_NCLS <= 32 && DP_NCLS == TELR_N_DPCLS, "class table");
    HIPCHK(hipMemcpyAsync(&tb_total, d_tboff + np, 8, hipMemcpyDeviceToHost, st));
    if (primary) HIPCHK(hipMemcpyAsync(&cg_total, d_cgoff + np, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_cls, d_clscnt, 128, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // one sort of (class, decreasing steps) keys gives every class list as a contiguous range of d_clslist
    coff.off[0] = 0;
    for (int c = 0; c < DP_NCLS; ++c) coff.off[c + 1] = coff.off[c] + h_cls[c];
    {
        size_t tbytes = 0;
        HIPCHK(rocprim::radix_sort_pairs(nullptr, tbytes, d_clskey, d_keytmp, d_listtmp, d_clslist, (size_t)np, 0, 24, st));
        void *tmp; TRY(ctx_buf(ctx, "rp_tmp", tbytes, &tmp));
        HIPCHK(rocprim::radix_sort_pairs(tmp, tbytes, d_clskey, d_keytmp, d_listtmp, d_clslist, (size_t)np, 0, 24, st));
    }
    if (np > 0) {       // trace-back pieces laid out in class-list order (page locality of the spill and of the walk)
        int64_t *d_tbs;
        TRY(ctx_buf_t(ctx, ("tb_sorted" + sfx).c_str(), (size_t)np + 1, &d_tbs));
        hipLaunchKernelGGL(k_tb_gather, dim3(np / 256 + 1), dim3(256), 0, st, d_tbb, d_clslist, np, coff, d_tbs);
        TRY((dev_exclusive_scan<int64_t, int64_t>(ctx, d_tbs, d_tboff, (size_t)np + 1)));
        hipLaunchKernelGGL(k_tb_scatter, dim3(np / 256 + 1), dim3(256), 0, st, d_probs, d_clslist, np, coff, d_tboff);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&tb_total, d_tboff + np, 8, hipMemcpyDeviceToHost, st));     // waves of interleaved problems are padded to their longest
        HIPCHK(hipStreamSynchronize(st));
    }
    uint8_t *d_tb;
    TRY(ctx_buf_t(ctx, ("tb" + sfx).c_str(), (size_t)tb_total + 256, &d_tb));
    if (primary) TRY(ctx_buf_t(ctx, "rawcig", (size_t)cg_total + 16, d_rawcig_io));
    DpArgs D; D.qseq2 = qs->d_seq2; D.qnmask = qs->d_nmask; D.tseq2 = tg->d_seq2; D.tnmask = tg->d_nmask; D.probs = d_probs;
    D.qtot = qs->padded_bases; D.ttot = tg->padded_bases;
    D.o.a = mo->a; D.o.b = mo->b; D.o.q = mo->q; D.o.e = mo->e; D.o.q2 = mo->q2; D.o.e2 = mo->e2; D.o.sc_ambi = mo->sc_ambi; D.o.zdrop = mo->zdrop;
    D.o.cx_scale = D.o.cx_open = D.o.cx_emax = D.o.cx_emin = D.o.cx_dec = D.o.cx_flat = 0;
    if (mo->cx_scale > 0) {          // convex gap cost: every score of the DP in 1/cx_scale units (k_chain_stats scales the records' sums back)
        const int S = mo->cx_scale;
        D.o.a *= S; D.o.b *= S; D.o.sc_ambi *= S; D.o.zdrop *= S;
        D.o.cx_scale = S; D.o.cx_open = mo->cx_open; D.o.cx_emax = mo->cx_ext_max; D.o.cx_emin = mo->cx_ext_min; D.o.cx_dec = mo->cx_decay;
        D.o.cx_flat = mo->cx_decay > 0 && mo->cx_ext_max > mo->cx_ext_min ? (mo->cx_ext_max - mo->cx_ext_min + mo->cx_decay - 1) / mo->cx_decay : 0;
    }
    D.tb = d_tb; D.cig = *d_rawcig_io; D.res = d_res; D.dcap = 0;
    D.retry = d_retry; D.tb4 = tb4_mask(mo); D.tag8_steps = tag8_steps(mo);
    static const int CAP[5] = { 64, 128, 256, 1024, DP_DMAX };
    // trace-back per class list, right behind the class's forward kernel on the same stream (TELR_AB=tb_one_launch: one
    // trace-back launch over all problems after every forward kernel has finished)
    static const bool tb_split = !ab_on("tb_one_launch");
    // wave table of the packed launch (all packed classes in one launch, waves ordered by decreasing cost); built
    // before the tail classes are started so that these two small launches do not queue behind them
    int nw = 0; uint32_t *d_wv2 = nullptr;
    {
        static const int LPP_[PK_NC] = { 1, 1, 1, 1, 2, 2, 2, 1, 4 };
        PkPlan plan; plan.woff[0] = 0;
        for (int c = 0; c < PK_NC; ++c) { const int ppw = 64 / LPP_[c]; plan.woff[c + 1] = plan.woff[c] + (h_cls[PK_CLS(c)] + ppw - 1) / ppw; }
        nw = plan.woff[PK_NC];
        if (nw > 0) {
            uint32_t *d_wk, *d_wv, *d_wk2;
            TRY(ctx_buf_t(ctx, ("pk_wkey" + sfx).c_str(), (size_t)nw, &d_wk));
            TRY(ctx_buf_t(ctx, ("pk_wval" + sfx).c_str(), (size_t)nw, &d_wv));
            TRY(ctx_buf_t(ctx, ("pk_wkey2" + sfx).c_str(), (size_t)nw, &d_wk2));
            TRY(ctx_buf_t(ctx, ("pk_wval2" + sfx).c_str(), (size_t)nw, &d_wv2));
            hipLaunchKernelGGL(k_pk_waves, dim3((nw + 255) / 256), dim3(256), 0, st, d_probs, d_clslist, coff, plan, d_wk, d_wv);
            HIPCHK(hipGetLastError());
            size_t tbytes = 0;
            HIPCHK(rocprim::radix_sort_pairs_desc(nullptr, tbytes, d_wk, d_wk2, d_wv, d_wv2, (size_t)nw, 0, 16, st));
            void *tmp; TRY(ctx_buf(ctx, "rp_tmp", tbytes, &tmp));
            HIPCHK(rocprim::radix_sort_pairs_desc(tmp, tbytes, d_wk, d_wk2, d_wv, d_wv2, (size_t)nw, 0, 16, st));
        }
    }
    // The few long/wide problems are latency-bound single waves: start each tail class on its own side
    // stream so that they run underneath the bulk classes on the main stream.
    HIPCHK(hipEventRecord(ctx->ev_fork, st));
    int side = 0;
    std::vector<hipStream_t> used;
    static const bool serial = getenv("TELR_SERIAL") != nullptr;      // profiling aid: every class on the main stream
    auto side_stream = [&]() { if (serial) return st; hipStream_t s2 = ctx->side[side % TELR_NSIDE]; ++side; used.push_back(s2); return s2; };
    // launch order = expected single-problem latency, longest first (the device offers only a few hardware queues,
    // so streams beyond that share one and run in submission order)
    static const int SIDE_ORDER[] = { 9, 4, 3, 21, 8, 20, 24, 23, 18, 2, 7, 19, 1, 0 };
    for (int c : SIDE_ORDER) {
        if (h_cls[c] == 0) continue;
        hipStream_t s2 = side_stream();
        HIPCHK(hipStreamWaitEvent(s2, ctx->ev_fork, 0));
        const int nl = h_cls[c];
        D.list = d_clslist + coff.off[c]; D.nlist = nl; D.dcap = 0;
        if (c <= 4) {
            D.dcap = CAP[c];
            size_t lds = (size_t)(CAP[c] + 2) * 5 * 4;
            // classes 3 and 4 (bands of 257 .. DP_DMAX diagonals: fills only) with four waves per problem (TELR_AB=dp_one_wave: one, as classes 0-2)
            static const bool one_wave = ab_on("dp_one_wave");
            if (c >= 3 && !one_wave) {
                if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k_dp_w4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(k_dp_w4, dim3(nl), dim3(256), lds, s2, D);
            } else {
                if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k_dp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(k_dp, dim3(nl), dim3(64), lds, s2, D);
            }
        }
        else if (c == 9) hipLaunchKernelGGL((k_dp_reg<256, 2, 4>), dim3(nl), dim3(256), 0, s2, D);
        else if (c == 8) hipLaunchKernelGGL((k_dp_reg<128, 2, 2>), dim3(nl), dim3(128), 0, s2, D);
        else if (c == 7) hipLaunchKernelGGL((k_dp_reg<64, 2>), dim3(nl), dim3(64), 0, s2, D);
        else if (c == 21) hipLaunchKernelGGL((k_dp_pkw<4>), dim3(nl), dim3(256), 0, s2, D);
        else if (c == 20) hipLaunchKernelGGL((k_dp_pkw<2>), dim3(nl), dim3(128), 0, s2, D);
        else if (c == 19) hipLaunchKernelGGL((k_dp_pkw<1>), dim3(nl), dim3(64), 0, s2, D);
        else if (c == 23) { const int ppw = 64 / PKX8_LPP; hipLaunchKernelGGL(k_dp_pkx_w8, dim3((nl + ppw - 1) / ppw), dim3(64), 0, s2, D); }
        else if (c == 24) hipLaunchKernelGGL(k_dp_pkx_w16, dim3((nl + 3) / 4), dim3(64), 0, s2, D);
        else if (nl < 8192) hipLaunchKernelGGL(k_dp_pkx16, dim3((nl + 3) / 4), dim3(64), 0, s2, D);       // few problems: latency counts
        else { const int ppw = 64 / PKX_LPP; hipLaunchKernelGGL(k_dp_pkx, dim3((nl + ppw - 1) / ppw), dim3(64), 0, s2, D); }
        HIPCHK(hipGetLastError());
        if (tb_split) {
            // few long problems: one wave walks one problem (latency); many (extensions, or a tail class with thousands of problems
            // on a repeat-rich genome): one lane per problem (throughput)
            const int tbw_max = 2048;
            if (c == 18 || c == 23 || c == 24 || c == 0 || nl > tbw_max) hipLaunchKernelGGL(k_traceback, dim3((nl + 63) / 64), dim3(64), 0, s2, d_probs, d_res, nl, d_tb, *d_rawcig_io, d_retry, D.list);
            else hipLaunchKernelGGL(k_traceback_w, dim3(nl), dim3(64), 0, s2, d_probs, d_res, nl, d_tb, *d_rawcig_io, d_retry, D.list);
            HIPCHK(hipGetLastError());
        }
    }
    D.dcap = 0;
    // packed classes: the wave table is cut into chunks; the trace-back of a chunk (memory bound) runs on its own
    // stream underneath the forward pass (issue bound) of the next chunk
    const int pk_chunks = 1;
    const bool tb_over = tb_split && !serial && nw > 0;
    if (primary) HIPCHK(hipEventRecord(ctx->evk[5], st));
    if (primary && nw > 0) ++ctx->pk_launches;
    if (nw > 0) {
        D.list = nullptr; D.nlist = 0;
        const int G = nw < 4096 ? 1 : pk_chunks;
        if (tb_over) { HIPCHK(hipStreamWaitEvent(ctx->tb_stream, ctx->ev_fork, 0)); if (primary) HIPCHK(hipEventRecord(ctx->evk[3], ctx->tb_stream)); }
        for (int g = 0; g < G; ++g) {
            const int w0 = (int)((int64_t)nw * g / G), w1 = (int)((int64_t)nw * (g + 1) / G);
            if (w1 <= w0) continue;
            hipLaunchKernelGGL(k_dp_pk, dim3(w1 - w0), dim3(64), 0, st, D, d_wv2 + w0, d_clslist, coff);
            HIPCHK(hipGetLastError());
            static const bool dbg_tb_skip = ab_on("dbg_tb_skip");        // EXPERIMENT ONLY (wrong records): what would the step be without the packed classes' trace-back?
            if (tb_over && !dbg_tb_skip) {
                HIPCHK(hipEventRecord(ctx->ev_chunk[g], st));
                HIPCHK(hipStreamWaitEvent(ctx->tb_stream, ctx->ev_chunk[g], 0));
                hipLaunchKernelGGL(k_traceback_pk, dim3(w1 - w0), dim3(64), 0, ctx->tb_stream, d_probs, d_res, d_tb, *d_rawcig_io, d_retry, d_wv2 + w0, d_clslist, coff, D.tb4, D.o, D.tag8_steps);
                HIPCHK(hipGetLastError());
            }
        }
        if (tb_over) { if (primary) HIPCHK(hipEventRecord(ctx->evk[4], ctx->tb_stream)); used.push_back(ctx->tb_stream); }
    }
    if (primary) HIPCHK(hipEventRecord(ctx->evk[0], st));
    if (h_cls[5]) { D.list = d_clslist + coff.off[5]; D.nlist = h_cls[5]; hipLaunchKernelGGL((k_dp_reg<32, 1>), dim3((h_cls[5] + 1) / 2), dim3(64), 0, st, D); }
    if (h_cls[6]) { D.list = d_clslist + coff.off[6]; D.nlist = h_cls[6]; hipLaunchKernelGGL((k_dp_reg<64, 1>), dim3(h_cls[6]), dim3(64), 0, st, D); }
    if (primary) HIPCHK(hipEventRecord(ctx->evk[1], st));
    if (tb_split) {
        if (!tb_over && primary) HIPCHK(hipEventRecord(ctx->evk[3], st));
        if (!tb_over && nw > 0) hipLaunchKernelGGL(k_traceback_pk, dim3(nw), dim3(64), 0, st, d_probs, d_res, d_tb, *d_rawcig_io, d_retry, d_wv2, d_clslist, coff, D.tb4, D.o, D.tag8_steps);
        for (int c = 6; c >= 5; --c) {
            if (h_cls[c] == 0) continue;
            hipLaunchKernelGGL(k_traceback, dim3((h_cls[c] + 63) / 64), dim3(64), 0, st, d_probs, d_res, h_cls[c], d_tb, *d_rawcig_io, d_retry, (const int32_t*)(d_clslist + coff.off[c]));
        }
        if (!tb_over && primary) HIPCHK(hipEventRecord(ctx->evk[4], st));
        HIPCHK(hipGetLastError());
    }
    for (size_t u = 0; u < used.size(); ++u) {
        HIPCHK(hipEventRecord(ctx->ev_side[u % TELR_NSIDE], used[u]));
        HIPCHK(hipStreamWaitEvent(st, ctx->ev_side[u % TELR_NSIDE], 0));
    }
#ifdef TB_PROF
    if (nw > 0) {
        HIPCHK(hipDeviceSynchronize());
        unsigned long long h[8]; HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tb_prof), sizeof(h)));
        unsigned long long z[8] = {0, 0, 0, 0, ~0ULL, 0, 0, 0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_tb_prof), z, sizeof(z)));
        if (h[2]) fprintf(stderr, "[tb prof] k_traceback_pk: %llu waves, span %.3f ms, longest wave %.3f ms (%llu lines), mean wave %.3f ms, %.1f lines per wave, wave-ms / span = %.1f waves on the device on average\n",
                          h[2], (double)(h[5] - h[4]) / 1e5, (double)(h[1] >> 16) / 1e5, h[1] & 0xffffULL, (double)h[0] / (double)h[2] / 1e5, (double)h[3] / (double)h[2], (double)h[0] / (double)(h[5] - h[4] ? h[5] - h[4] : 1));
    }
#endif
    if (!tb_split) {
        if (primary) HIPCHK(hipEventRecord(ctx->evk[3], st));
        hipLaunchKernelGGL(k_traceback, dim3((np + 63) / 64), dim3(64), 0, st, d_probs, d_res, np, d_tb, *d_rawcig_io, d_retry, (const int32_t*)nullptr);
        if (primary) HIPCHK(hipEventRecord(ctx->evk[4], st));
    }
    HIPCHK(hipGetLastError());
    return TELR_OK;
}

// TELR_TRACE=host: wall-clock marks of the host side of every batch on stderr (where does a range wait for the host?)
struct HostTrace {
    bool on; const char *tag; std::chrono::steady_clock::time_point t0, last;
    HostTrace(const char *t) : tag(t) { static const bool e = trace_on("host"); on = e; t0 = last = std::chrono::steady_clock::now(); }
    void mark(const char *what) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[host %s] %-28s +%7.3f ms  (%8.3f)\n", tag, what, std::chrono::duration<double, std::milli>(now - last).count(), std::chrono::duration<double, std::milli>(now - t0).count());
        last = now;
    }
};

// ---------------------------------------------------------------------------------------
// one batch of queries [q0, q1)
struct OccCut { int32_t mid_occ; const int32_t *d_tmid; };      // pooled cut-off; per-target cut-offs (nullable)
static int map_batch(telr_ctx *ctx, const telr_index *ix, const telr_seqset *qs, const int32_t *d_qtarget, int32_t q0, int32_t q1,
                     const telr_map_opt *mo, OccCut occ, telr_result *R, RangeTurn *gate = nullptr)
{
    const int32_t mid_occ = occ.mid_occ;
    const int nq = q1 - q0, k = ix->io.k, w = ix->io.w;
    const telr_seqset *tg = ix->targets;
    hipStream_t st = ctx->stream;
    HostTrace ht("batch");

    // ---- sketch -------------------------------------------------------------------------
    // queries in descending length order for the one-block-per-query kernels (their tail is the longest read): sorted by
    // a helper thread while the sketch kernels run, uploaded from pinned memory before the seeding stage
    int32_t *d_qorder, *h_ord;
    TRY(ctx_hbuf_t(ctx, "h_qorder", (size_t)nq + 1, &h_ord));
    TRY(ctx_buf_t(ctx, "q_order", (size_t)nq + 1, &d_qorder));
    std::thread ord_thread([&]() {
        for (int i = 0; i < nq; ++i) h_ord[i] = i;
        std::stable_sort(h_ord, h_ord + nq, [&](int32_t x, int32_t y) { return qs->len[q0 + x] > qs->len[q0 + y]; });
    });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } ord_join{ord_thread};     // error paths return early
    StageTimer t_sk(ctx, ST_SKETCH, true);
    TileList &T = ctx_tiles(ctx);
    uint64_t *d_mx; uint32_t *d_my; int32_t *d_toff; int32_t nmz = 0;
    // the seeding kernels read the sketch kernel's staging arrays in place (no compaction copy); TELR_AB=mz_compact for A/B
    static const bool mz_compact_env = ab_on("mz_compact");
    // sub-read voting (spec 3.10) applies to all-vs-all calls only; its kernel reads the compacted minimizer arrays
    const bool vote = mo->vote_len > 0 && !d_qtarget && !(mo->flags & TELR_MF_PER_TARGET);
    const bool mz_staged = !ix->io.is_hpc && !mz_compact_env && !vote;
    if (ix->io.is_hpc) {
        SketchHpcArgs H; std::vector<int32_t> nrun;
        TRY(build_hpc(ctx, qs, q0, q1, "qh_", &H, &nrun));
        make_tiles_f(q0, q1, q0, [&](int32_t q) { return nrun[q - q0] - k + 1; }, T);
        TRY(run_sketch(ctx, qs, T, k, w, nullptr, "q_", &d_mx, &d_my, &d_toff, &nmz, &H));
    } else {
        make_tiles(qs, q0, q1, k, T);
        TRY(run_sketch(ctx, qs, T, k, w, nullptr, "q_", &d_mx, &d_my, &d_toff, &nmz, nullptr, mz_staged));
    }
    // per-query minimizer offsets = tile_off[first tile of the query]
    int32_t *d_first, *d_qmz;
    TRY(ctx_buf_t(ctx, "q_first", (size_t)nq + 1, &d_first));
    TRY(ctx_buf_t(ctx, "q_mzoff", (size_t)nq + 1, &d_qmz));
    HIPCHK(hipMemcpyAsync(d_first, T.first.data(), (size_t)nq * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_first + nq, &T.n, 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gather_i32, dim3((nq + 256) / 256), dim3(256), 0, st, d_toff, d_first, nq, nmz, d_qmz);
    HIPCHK(hipGetLastError());
    t_sk.stop(); ht.mark("sketch issued");

    ord_thread.join();
    HIPCHK(hipMemcpyAsync(d_qorder, h_ord, (size_t)nq * 4, hipMemcpyHostToDevice, st));

    // ---- seeding --------------------------------------------------------------------------
    StageTimer t_sd(ctx, ST_SEED, true);
    IndexView I; I.ent_hash = ix->d_ent_hash; I.ent_off = ix->d_ent_off; I.pos = ix->d_pos; I.bstart = nullptr; I.goff = ix->d_goff;
    I.tlen = tg->d_len; I.n_ent = ix->n_ent; I.shift = ix->shift; I.k = k; I.w = w;
    I.ht = ix->d_ht; I.ht_shift = ix->ht_shift; I.ht_mask = ix->ht_mask; I.ht_home = ix->d_ht_home;
    int32_t *d_mcnt, *d_maoff, *d_qaoff;
    int32_t *d_ment, *d_mn;
    TRY(ctx_buf_t(ctx, "mz_ent", (size_t)nmz + 1, &d_ment));
    TRY(ctx_buf_t(ctx, "mz_n", (size_t)nmz + 1, &d_mn));
    TRY(ctx_buf_t(ctx, "mz_cnt", (size_t)nmz + 1, &d_mcnt));
    TRY(ctx_buf_t(ctx, "mz_aoff", (size_t)nmz + 1, &d_maoff));
    TRY(ctx_buf_t(ctx, "q_aoff", (size_t)nq + 1, &d_qaoff));
    SeedArgs S; S.I = I; S.mz_x = d_mx; S.mz_y = d_my; S.q_mzoff = d_qmz; S.qlen = qs->d_len + q0; S.qtarget = d_qtarget ? d_qtarget + q0 : nullptr;
    S.mid_occ = mid_occ; S.tmid = occ.d_tmid; S.per_target = (mo->flags & TELR_MF_PER_TARGET) ? 1 : 0; S.n_targets = tg->n; S.mz_cnt = d_mcnt; S.mz_ent = d_ment; S.mz_n = d_mn; S.mz_aoff = nullptr; S.q_cnt = nullptr; S.q_aoff = nullptr; S.lds_keys = nullptr; S.keys = nullptr; S.q_order = d_qorder;
    S.tile_off = mz_staged ? d_toff : nullptr; S.q_tile0 = mz_staged ? d_first : nullptr;
    VoteOpt VO; VO.len = mo->vote_len; VO.shift = mo->vote_bin_shift; VO.vmin = mo->vote_min; VO.frac_q8 = mo->vote_frac_q8;
    VoteArgs VA; memset(&VA, 0, sizeof(VA));
    // the per-query anchor counts: from the seeding kernel (which also leaves every minimizer's offset inside its query), or from the
    // vote kernel when the sub-reads vote (it appends a query's survivors to its piece of a staging array; k_vote_compact moves them to
    // the scanned offsets)
    int32_t *d_qcnt;
    TRY(ctx_buf_t(ctx, "q_cnt", (size_t)nq + 2, &d_qcnt));
    S.q_cnt = d_qcnt; S.mz_aoff = d_maoff; S.q_aoff = d_qaoff;
    int64_t *d_qsoff = nullptr; uint64_t *d_stage = nullptr;
    int64_t *d_na64; TRY(ctx_buf_t(ctx, "seed_na64", 2, &d_na64));
    int32_t *d_nover = (int32_t*)(d_na64 + 1);
    if (vote) {
        int64_t *d_qhits;
        TRY(ctx_buf_t(ctx, "vote_qhits", (size_t)nq + 2, &d_qhits)); TRY(ctx_buf_t(ctx, "vote_qsoff", (size_t)nq + 2, &d_qsoff));
        if (nmz) {
            static const bool always_filter = ab_on("vote_filter");
            if (k <= 13 && !always_filter) hipLaunchKernelGGL(k_vote_lookup<false>, dim3((unsigned)((nmz + 255) / 256)), dim3(256), 0, st, I, d_mx, nmz, mid_occ, d_ment, d_mn);
            else hipLaunchKernelGGL(k_vote_lookup<true>, dim3((unsigned)((nmz + 255) / 256)), dim3(256), 0, st, I, d_mx, nmz, mid_occ, d_ment, d_mn);
        }
        hipLaunchKernelGGL(k_vote_qhits, dim3(nq + 1), dim3(64), 0, st, d_qmz, d_mn, nq, d_qhits);
        TRY((dev_qscan<int64_t, int64_t>(ctx, d_qhits, nq, d_qsoff, nullptr, 0, nullptr, nullptr)));
        HIPCHK(hipGetLastError());
        int64_t nhits = 0;
        HIPCHK(hipMemcpyAsync(&nhits, d_qsoff + nq, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        TRY(ctx_buf_t(ctx, "vote_stage", (size_t)nhits + 1, &d_stage));
        VA.q_soff = d_qsoff; VA.stage = d_stage; VA.q_cnt = d_qcnt;
        // queries whose hits fit 16-bit vote counters (nearly all) with the half-size table, the others with 32-bit counters
        const int64_t lim16 = 65535;
        hipLaunchKernelGGL(k_seed_vote<true>, dim3(nq), dim3(64 * VOTE_WAVES), 0, st, S, VO, VA, std::min<int64_t>(lim16, 65535));
        hipLaunchKernelGGL(k_seed_vote<false>, dim3(nq), dim3(64 * VOTE_WAVES), 0, st, S, VO, VA, std::min<int64_t>(lim16, 65535));
    } else hipLaunchKernelGGL(k_seed<0>, dim3(nq), dim3(256), 0, st, S);
    HIPCHK(hipGetLastError());
    // per-query anchor offsets (ONE workgroup: k_qscan), the anchor total in 64 bits -- the anchors of a batch are addressed with int32
    // offsets, a batch with 2^31 anchors or more is handed back to the caller, which halves it -- and how many queries hold more anchors
    // than one workgroup sorts in LDS (segsort.hip.h)
    int32_t *d_overlist; TRY(ctx_buf_t(ctx, "q_overlist", (size_t)nq + 1, &d_overlist));
    TRY((dev_qscan<int32_t, int32_t>(ctx, d_qcnt, nq, d_qaoff, d_na64, (int64_t)SEGSORT_CAP, d_nover, d_overlist)));
    HIPCHK(hipGetLastError());
    int32_t na = 0; int64_t na64 = 0; int32_t n_over = 0;
    HIPCHK(hipMemcpyAsync(&na, d_qaoff + nq, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&na64, d_na64, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&n_over, d_nover, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (na64 >= (1LL << 31) - 256) { stage_collect(ctx); return TELR_SPLIT_RANGE; }
    ctx->ctr.minimizers += nmz; ctx->ctr.probes += nmz;
    ctx->ctr.over_queries += n_over; ctx->ctr.over_ranges += n_over > 0;
    static const bool lib_sort = ab_on("sort64");      // A/B: rocPRIM's segmented radix sort for every query
    const bool any_over = n_over > 0;
    uint64_t *d_keys, *d_skeys;
    static const bool seed_unfused_env = ab_on("seed_unfused");
    TRY(ctx_buf_t(ctx, "keys", (!lib_sort && !any_over && (vote || !seed_unfused_env)) ? (size_t)1 : (size_t)na, &d_keys));       // only the two-step forms need the unsorted keys in memory
    TRY(ctx_buf_t(ctx, "skeys", (size_t)na, &d_skeys));
    S.keys = d_keys;
    // sub-read voting: the LDS sort reads the survivors from the staging pieces in place; only the library sort needs them dense
    // A range that holds an over-size query takes the two-step form as a whole (the round-5 rule).  TELR_AB=over_routed: only the over-size
    // queries do (their keys written on a side stream under the LDS sort of the others) -- built in round 6 for the hard genome, where EVERY
    // range holds such a read, and measured SLOWER on configs[2] (196.7 against 185.0 ms per step, same box, three alternating runs each:
    // profiles/r06_oversize_routing_ab.txt) and no faster on the hard genome; kept behind the switch, tests/test_gpu_switches.py.
    static const bool over_routed = ab_on("over_routed");
    const bool over_whole = any_over && !over_routed;
    const bool vote_in_place = vote && !lib_sort && !over_whole;
    // The anchor keys are MADE inside the sort (SeedProducer: the seeding routine writes a query's keys straight into the sorting
    // workgroup's LDS), so unsorted keys never exist in HBM; TELR_AB=seed_unfused keeps the two-step form for A/B, and a range with a
    // query above the LDS limit takes it too (its library sort reads the keys from memory)
    static const bool seed_unfused = ab_on("seed_unfused");
    const bool seed_fused = !vote && !lib_sort && !seed_unfused && !over_whole;
    S.lds_keys = nullptr;
    // (TELR_AB=over_routed: the over-size queries' keys go to memory on a SIDE stream, under the LDS sort of everybody else; the library's sort waits for it)
    hipEvent_t over_ready = nullptr;
    if (vote) {
        if (!vote_in_place) hipLaunchKernelGGL(k_vote_compact, dim3(nq), dim3(256), 0, st, d_stage, d_qsoff, d_qaoff, nq, d_keys, (const int32_t*)nullptr);
        else if (any_over) {
            HIPCHK(hipEventRecord(ctx->ev_fork, st)); HIPCHK(hipStreamWaitEvent(ctx->side[0], ctx->ev_fork, 0));
            hipLaunchKernelGGL(k_vote_compact, dim3(n_over), dim3(256), 0, ctx->side[0], d_stage, d_qsoff, d_qaoff, n_over, d_keys, (const int32_t*)d_overlist);
            HIPCHK(hipEventRecord(ctx->ev_side[0], ctx->side[0])); over_ready = ctx->ev_side[0];
        }
    }
    else if (!seed_fused) hipLaunchKernelGGL(k_seed<1>, dim3(nq), dim3(256), 0, st, S);
    else if (any_over) {
        SeedArgs So = S; So.q_order = d_overlist;
        HIPCHK(hipEventRecord(ctx->ev_fork, st)); HIPCHK(hipStreamWaitEvent(ctx->side[0], ctx->ev_fork, 0));
        hipLaunchKernelGGL(k_seed<1>, dim3(n_over), dim3(256), 0, ctx->side[0], So);
        HIPCHK(hipEventRecord(ctx->ev_side[0], ctx->side[0])); over_ready = ctx->ev_side[0];
    }
    HIPCHK(hipGetLastError());
    t_sd.stop(); ht.mark("seed (sync: anchor total)");
    ctx->ctr.anchors += na;

    // ---- per-query sort of the anchor keys --------------------------------------------------
    StageTimer t_so(ctx, ST_SORT, true);
    if (na > 0 && seed_fused) { SeedProducer sp; sp.S = S; TRY((seg_sort_u64<SeedProducer>(ctx, "so_a", nullptr, d_skeys, d_qaoff, d_qaoff + 1, nullptr, d_qorder, nq, (size_t)na, any_over, st, sp, d_keys, over_ready))); }
    else if (na > 0) TRY(seg_sort_u64(ctx, "so_a", vote_in_place ? d_stage : d_keys, d_skeys, d_qaoff, d_qaoff + 1, vote_in_place ? d_qsoff : nullptr, d_qorder, nq, (size_t)na, any_over, st, LoadKeys(), d_keys, over_ready));
    t_so.stop();

    // ---- chaining ---------------------------------------------------------------------------
    StageTimer t_ch(ctx, ST_CHAIN, true);
    int32_t *d_f, *d_p;
    TRY(ctx_buf_t(ctx, "chain_f", (size_t)na, &d_f));
    TRY(ctx_buf_t(ctx, "chain_p", (size_t)na, &d_p));
    // long join: anchors are chained within max(bw, bw_long) diagonals
    ChainOpt co; co.max_gap = mo->max_gap; co.bw = mo->bw_long > mo->bw ? mo->bw_long : mo->bw; co.min_cnt = mo->min_cnt; co.min_chain_score = mo->min_chain_score;
    co.chain_gap_q8 = mo->chain_gap_q8; co.chain_skip_q8 = mo->chain_skip_q8;
    // which loop: chosen per run of anchors (kernels.hip.h: WHICH LOOP); A/B: chain_push = the full push loop (every one of the H links scored)
    // everywhere, chain_lazy = the lazy far look-back everywhere, chain_no_mw = no second kernel for the long dense runs.
    // TELR_CHAIN_DENSE=n,span,mw_n moves the choices (tests, experiments).
    static const bool chain_push = ab_on("chain_push"), chain_lazy = ab_on("chain_lazy"), chain_no_mw = ab_on("chain_no_mw");
    static int dense_v[3] = {2048, 48, 1 << 19};
    static const bool dense_env = [] { if (const char *e = getenv("TELR_CHAIN_DENSE")) sscanf(e, "%d%*[,:]%d%*[,:]%d", &dense_v[0], &dense_v[1], &dense_v[2]); return true; }();
    (void)dense_env;
    const int Rr = mo->chain_lookback / 64;
    const bool mw_all = dense_v[2] <= SEGSORT_CAP;            // (small thresholds: every query is looked at, not only the over-size list)
    static const bool no_islands = ab_on("no_islands");       // A/B: one wave per query whatever the call's shape
    const bool islands = nq <= CHAIN_ISL_NQ && na > 0 && !no_islands;
    const bool mw = !chain_push && !chain_lazy && !chain_no_mw && Rr >= 2 && na > 0 && !islands && dense_v[2] > 0 && (mw_all || n_over > 0);
    co.dense_n = dense_v[0]; co.dense_span = dense_v[1]; co.mw_n = mw ? dense_v[2] : 0;
#define CHAIN_LAUNCH(RR, SK) do { if (chain_push) hipLaunchKernelGGL((k_chain<RR, SK, 0>), dim3(nq), dim3(64), 0, st, d_skeys, d_qaoff, nq, co, d_f, d_p, d_qorder, 0); \
                                  else if (chain_lazy) hipLaunchKernelGGL((k_chain<RR, SK, 1>), dim3(nq), dim3(64), 0, st, d_skeys, d_qaoff, nq, co, d_f, d_p, d_qorder, 0); \
                                  else hipLaunchKernelGGL((k_chain<RR, SK, 2>), dim3(nq), dim3(64), 0, st, d_skeys, d_qaoff, nq, co, d_f, d_p, d_qorder, mw ? 1 : 0); } while (0)
#define CHAIN_MW(RR, SK) hipLaunchKernelGGL((k_chain_mw<RR, SK>), dim3(mw_all ? nq : n_over), dim3(64 * RR), 0, ctx->side[0], d_skeys, d_qaoff, mw_all ? nq : n_over, mw_all ? (const int32_t*)nullptr : (const int32_t*)d_overlist, co, d_f, d_p)
    const bool skip = co.chain_skip_q8 != 0;
    if (islands) {
        // few queries: chain island by island (kernels.hip.h: ISLANDS) -- same f and p, thousands of waves instead of nq
        int32_t *d_head, *d_rank, *d_ioff, *d_ipd;
        TRY(ctx_buf_t(ctx, "isl_head", (size_t)na + 1, &d_head));
        TRY(ctx_buf_t(ctx, "isl_rank", (size_t)na + 1, &d_rank));
        TRY(ctx_buf_t(ctx, "isl_off", (size_t)na + 2, &d_ioff));
        TRY(ctx_buf_t(ctx, "isl_pd", (size_t)na + 1, &d_ipd));
        HIPCHK(hipMemsetAsync(d_head + na, 0, 4, st));
        hipLaunchKernelGGL(k_isl_heads, dim3(nq), dim3(256), 0, st, d_skeys, d_qaoff, nq, (uint32_t)co.max_gap, d_head);
        HIPCHK(hipGetLastError());
        TRY((dev_exclusive_scan<int32_t, int32_t>(ctx, d_head, d_rank, (size_t)na + 1)));
        hipLaunchKernelGGL(k_isl_fill, dim3(nq), dim3(256), 0, st, d_qaoff, nq, d_head, d_rank, (int32_t)na, d_ioff, d_ipd);
        const unsigned grid = (unsigned)std::min<int64_t>((int64_t)na, 256 * 32);
#define CHAIN_ISL(RR, SK) do { if (chain_push) hipLaunchKernelGGL((k_chain_isl<RR, SK, 0>), dim3(grid), dim3(64), 0, st, d_skeys, d_ioff, d_ipd, d_rank + na, co, d_f, d_p); \
                               else if (chain_lazy) hipLaunchKernelGGL((k_chain_isl<RR, SK, 1>), dim3(grid), dim3(64), 0, st, d_skeys, d_ioff, d_ipd, d_rank + na, co, d_f, d_p); \
                               else hipLaunchKernelGGL((k_chain_isl<RR, SK, 2>), dim3(grid), dim3(64), 0, st, d_skeys, d_ioff, d_ipd, d_rank + na, co, d_f, d_p); } while (0)
        if (Rr == 1) { if (skip) CHAIN_ISL(1, true); else CHAIN_ISL(1, false); }
        else if (Rr == 2) { if (skip) CHAIN_ISL(2, true); else CHAIN_ISL(2, false); }
        else { if (skip) CHAIN_ISL(4, true); else CHAIN_ISL(4, false); }
#undef CHAIN_ISL
    }
    else {
        if (mw) {
            // the long dense runs beside the others: forked here, joined behind k_chain
            HIPCHK(hipEventRecord(ctx->ev_side[0], st));
            HIPCHK(hipStreamWaitEvent(ctx->side[0], ctx->ev_side[0], 0));
            if (Rr == 2) { if (skip) CHAIN_MW(2, true); else CHAIN_MW(2, false); }
            else { if (skip) CHAIN_MW(4, true); else CHAIN_MW(4, false); }
            HIPCHK(hipEventRecord(ctx->ev_side[0], ctx->side[0]));
        }
        if (Rr == 1) { if (skip) CHAIN_LAUNCH(1, true); else CHAIN_LAUNCH(1, false); }
        else if (Rr == 2) { if (skip) CHAIN_LAUNCH(2, true); else CHAIN_LAUNCH(2, false); }
        else { if (skip) CHAIN_LAUNCH(4, true); else CHAIN_LAUNCH(4, false); }
        if (mw) HIPCHK(hipStreamWaitEvent(st, ctx->ev_side[0], 0));
    }
#undef CHAIN_LAUNCH
#undef CHAIN_MW
    HIPCHK(hipGetLastError());
    t_ch.stop();

    // ---- peaks + back-tracking ----------------------------------------------------------------
    StageTimer t_bt(ctx, ST_BACKTRACK, true);
    uint8_t *d_flags; uint64_t *d_pk, *d_pk2, *d_canch; int32_t *d_npk, *d_pkend, *d_choff, *d_nch;
    TRY(ctx_buf_t(ctx, "flags", (size_t)na + 16, &d_flags));
    uint8_t *d_nonpeak = d_flags;
    TRY(ctx_buf_t(ctx, "pk", (size_t)na, &d_pk));
    TRY(ctx_buf_t(ctx, "pk2", (size_t)na, &d_pk2));
    TRY(ctx_buf_t(ctx, "canch", (size_t)na, &d_canch));
    TRY(ctx_buf_t(ctx, "n_peaks", (size_t)nq + 1, &d_npk));
    TRY(ctx_buf_t(ctx, "pk_end", (size_t)nq + 1, &d_pkend));
    TRY(ctx_buf_t(ctx, "ch_off", (size_t)nq + 1, &d_choff));
    TRY(ctx_buf_t(ctx, "n_chains", (size_t)nq + 1, &d_nch));
    HIPCHK(hipMemsetAsync(d_flags, 0, (size_t)na + 16, st));
    hipLaunchKernelGGL(k_nonpeak, dim3(nq), dim3(256), 0, st, d_qaoff, d_f, d_p, d_nonpeak);
    hipLaunchKernelGGL(k_peaks, dim3(nq), dim3(256), 0, st, d_qaoff, d_f, d_nonpeak, mo->min_chain_score, d_pk, d_npk, d_pkend);
    HIPCHK(hipGetLastError());
    if (na > 0) TRY(seg_sort_u64(ctx, "so_p", d_pk, d_pk2, d_qaoff, d_pkend, nullptr, d_qorder, nq, (size_t)na, any_over, st));
    HIPCHK(hipMemsetAsync(d_npk + nq, 0, 4, st));
    TRY((dev_exclusive_scan<int32_t, int32_t>(ctx, d_npk, d_choff, (size_t)nq + 1)));
    int32_t npk_tot = 0;
    HIPCHK(hipMemcpyAsync(&npk_tot, d_choff + nq, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    ChainRec *d_rec;
    TRY(ctx_buf_t(ctx, "chain_rec", (size_t)npk_tot, &d_rec));
    {
        // owner / depth sweeps (kernels.hip.h, "back-tracking without a walker"): five launches over all queries, longest first
        uint32_t *d_owner; int32_t *d_depth, *d_chtop, *d_chaoff;
        TRY(ctx_buf_t(ctx, "bt_owner", (size_t)na + 1, &d_owner));
        TRY(ctx_buf_t(ctx, "bt_depth", (size_t)na + 1, &d_depth));
        TRY(ctx_buf_t(ctx, "bt_chtop", (size_t)na + 1, &d_chtop));
        TRY(ctx_buf_t(ctx, "bt_chaoff", (size_t)na + 1, &d_chaoff));
        HIPCHK(hipMemsetAsync(d_owner, 0xff, ((size_t)na + 1) * 4, st));
        hipLaunchKernelGGL(k_bt_rank, dim3(nq), dim3(256), 0, st, d_qaoff, d_pk2, d_npk, d_owner);
        hipLaunchKernelGGL(k_bt_owner, dim3(nq), dim3(64), 0, st, d_qaoff, nq, d_p, mo->chain_lookback, d_owner, d_qorder);
        hipLaunchKernelGGL(k_bt_depth, dim3(nq), dim3(64), 0, st, d_qaoff, nq, d_p, d_owner, d_depth, d_chtop, d_qorder);
        hipLaunchKernelGGL(k_bt_emit, dim3(nq), dim3(64), 0, st, d_skeys, d_qaoff, nq, d_f, d_p, d_pk2, d_npk, d_choff, mo->min_chain_score, mo->min_cnt,
                           d_owner, d_depth, d_chtop, d_chaoff, d_rec, d_nch, d_qorder);
        hipLaunchKernelGGL(k_bt_scatter, dim3(nq), dim3(256), 0, st, d_skeys, d_qaoff, d_owner, d_depth, d_chaoff, d_canch, d_qorder);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipGetLastError());
    // Pass-1 chain selection runs on the device (k_select1).  The debug taps of the parity tests need every chain record on the host.
    const bool need_recs = ctx->debug != 0;
    int32_t *h_nch, *h_choff, *h_qaoff; ChainRec *h_rec;
    TRY(ctx_hbuf_t(ctx, "h_nch", (size_t)nq + 1, &h_nch));
    TRY(ctx_hbuf_t(ctx, "h_choff", (size_t)nq + 1, &h_choff));
    TRY(ctx_hbuf_t(ctx, "h_qaoff", (size_t)nq + 1, &h_qaoff));
    TRY(ctx_hbuf_t(ctx, "h_rec", need_recs ? (size_t)npk_tot + 1 : 1, &h_rec));
    if (need_recs) {
        HIPCHK(hipMemcpyAsync(h_nch, d_nch, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(h_choff, d_choff, (size_t)(nq + 1) * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(h_qaoff, d_qaoff, (size_t)(nq + 1) * 4, hipMemcpyDeviceToHost, st));
        if (npk_tot) HIPCHK(hipMemcpyAsync(h_rec, d_rec, (size_t)npk_tot * sizeof(ChainRec), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    t_bt.stop(); ht.mark("sort+chain+backtrack issued");
    ctx->dbg_na = na; ctx->dbg_nq = nq;

    // ---- chain boxes + selection pass 1 -> the kept chains (query-major, pass-1 rank order) ------------------------
    const int NT = host_threads();
    std::vector<int32_t> q_k0(nq + 1, 0);
    KeptLite *hl = nullptr;                  // what the host needs of the kept chains
    KeptChain *d_kc = nullptr;               // their descriptors for the problem builder (device)
    int nk = 0;
    int64_t n_chain_tot = 0;
    {
        StageTimer t_sel(ctx, ST_SELECT, true);
        uint64_t *d_sk, *d_sk2; int32_t *d_segend, *d_pfs, *d_pfe, *d_ptid, *d_pkey, *d_tct, *d_tcn, *d_nkept, *d_koff; uint8_t *d_keep;
        TRY(ctx_buf_t(ctx, "sel_key", (size_t)npk_tot + 1, &d_sk));
        TRY(ctx_buf_t(ctx, "sel_key2", (size_t)npk_tot + 1, &d_sk2));
        TRY(ctx_buf_t(ctx, "sel_segend", (size_t)nq + 1, &d_segend));
        TRY(ctx_buf_t(ctx, "sel_pfs", (size_t)npk_tot + 1, &d_pfs));
        TRY(ctx_buf_t(ctx, "sel_pfe", (size_t)npk_tot + 1, &d_pfe));
        TRY(ctx_buf_t(ctx, "sel_ptid", (size_t)npk_tot + 1, &d_ptid));
        TRY(ctx_buf_t(ctx, "sel_pkey", (size_t)npk_tot + 1, &d_pkey));
        TRY(ctx_buf_t(ctx, "sel_tct", (size_t)npk_tot + 1, &d_tct));
        TRY(ctx_buf_t(ctx, "sel_tcn", (size_t)npk_tot + 1, &d_tcn));
        TRY(ctx_buf_t(ctx, "sel_keep", (size_t)npk_tot + 1, &d_keep));
        TRY(ctx_buf_t(ctx, "sel_nkept", (size_t)nq + 2, &d_nkept));
        TRY(ctx_buf_t(ctx, "sel_koff", (size_t)nq + 2, &d_koff));
        hipLaunchKernelGGL(k_sel_keys, dim3(nq), dim3(64), 0, st, d_choff, d_nch, d_rec, d_sk, d_segend);
        HIPCHK(hipGetLastError());
        if (npk_tot > 0) TRY(seg_sort_u64(ctx, "so_c", d_sk, d_sk2, d_choff, d_segend, nullptr, d_qorder, nq, (size_t)npk_tot, any_over, st));
        SelOpt so; so.mask_level = mo->mask_level; so.pri_ratio = mo->pri_ratio; so.best_n = mo->best_n; so.secondary = mo->secondary;
        so.per_target = (mo->flags & TELR_MF_PER_TARGET) ? 1 : 0;
        hipLaunchKernelGGL(k_select1, dim3(nq), dim3(64), 0, st, d_choff, d_nch, d_rec, d_sk2, qs->d_len + q0, ix->d_goff, tg->n, so,
                           d_pfs, d_pfe, d_ptid, d_pkey, d_tct, d_tcn, d_keep, d_nkept, d_qorder);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemsetAsync(d_nkept + nq, 0, 4, st));
        TRY((dev_exclusive_scan<int32_t, int32_t>(ctx, d_nkept, d_koff, (size_t)nq + 1)));
        HIPCHK(hipMemcpyAsync(q_k0.data(), d_koff, (size_t)(nq + 1) * 4, hipMemcpyDeviceToHost, st));
        if (!need_recs) HIPCHK(hipMemcpyAsync(h_nch, d_nch, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        nk = q_k0[nq];
        for (int q = 0; q < nq; ++q) n_chain_tot += h_nch[q];
        KeptLite *d_kl;
        TRY(ctx_buf_t(ctx, "kept", (size_t)nk, &d_kc));
        TRY(ctx_buf_t(ctx, "kept_lite", (size_t)nk, &d_kl));
        TRY(ctx_hbuf_t(ctx, "h_kept_lite", (size_t)nk, &hl));
        if (nk > 0) {
            hipLaunchKernelGGL(k_select1_write, dim3(nq), dim3(64), 0, st, d_choff, d_nch, d_rec, d_sk2, d_keep, d_koff, d_qaoff, q0, qs->d_len + q0, qs->d_boff + q0,
                               ix->d_goff, tg->d_len, tg->d_boff, tg->n, d_kc, d_kl);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(hl, d_kl, (size_t)nk * sizeof(KeptLite), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
        }
        t_sel.stop();
        if (ctx->debug) {                    // every chain with its box, in discovery order (stage-level parity tests)
            ctx->dbg_chain.clear();
            for (int q = 0; q < nq; ++q) for (int c = 0; c < h_nch[q]; ++c) {
                const ChainRec &r = h_rec[h_choff[q] + c];
                const uint32_t g0 = (uint32_t)A_G(r.a0);
                const int tid = (int)(std::upper_bound(ix->goff.begin(), ix->goff.begin() + tg->n, g0) - ix->goff.begin()) - 1, go = (int)ix->goff[tid];
                int32_t v[9] = { q0 + q, r.score, r.cnt, (int)(r.a0 >> 63), tid, std::max(0, A_G(r.a0) - go - A_SPAN(r.a0) + 1), A_G(r.a1) - go + 1, A_Q(r.a0) - A_SPAN(r.a0) + 1, A_Q(r.a1) + 1 };      // (start clamped at the target's first base, as the kept chains are: k_select1_write)
                ctx->dbg_chain.insert(ctx->dbg_chain.end(), v, v + 9);
            }
        }
    }
    ctx->ctr.chains += n_chain_tot; ht.mark("selection (sync: kept chains)");

    // results per kept chain (chain-level numbers; overwritten by the DP numbers below)
    if (ctx->h_kal.size() < (size_t)nk) ctx->h_kal.resize((size_t)nk + (size_t)nk / 8);
    telr_aln *kal = ctx->h_kal.data();
    parallel_ranges(NT, nk, [&](int, int xa, int xb) {
        for (int x = xa; x < xb; ++x) {
            const KeptLite &c = hl[x];
            telr_aln &r = kal[x]; memset(&r, 0, sizeof(r));
            const int qlen = qs->len[c.qid];
            r.qid = c.qid; r.tid = c.tid; r.qlen = qlen; r.tlen = tg->len[c.tid]; r.score = c.score; r.cnt = c.cnt; r.flags = c.rev ? TELR_F_REV : 0;
            r.ts = c.rs; r.te = c.re;
            if (c.rev) { r.qs = qlen - c.qe; r.qe = qlen - c.qs; } else { r.qs = c.qs; r.qe = c.qe; }
            r.mlen = std::min(c.score, c.qe - c.qs); r.blen = std::max(c.qe - c.qs, c.re - c.rs); r.dp_score = c.score;
        }
    });

    const bool do_dp = (mo->flags & TELR_MF_CIGAR) && nk > 0;
    std::vector<int32_t> &h_poff = ctx->h_poffv;
    int64_t *h_foff = nullptr; size_t cig_base = 0; ChainStat *h_cs = nullptr; unsigned long long *h_acc = nullptr;
    int np = 0;
    if (do_dp) {
        // ---- DP problem list ----------------------------------------------------------------
        StageTimer t_sg(ctx, ST_SEGMENTS, true);
        int32_t *d_nprob, *d_poff;             // (the kept-chain descriptors d_kc are on the device already)
        TRY(ctx_buf_t(ctx, "nprob", (size_t)nk + 1, &d_nprob));
        TRY(ctx_buf_t(ctx, "prob_off", (size_t)nk + 1, &d_poff));
        hipLaunchKernelGGL(k_segments_w<0>, dim3(nk), dim3(64), 0, st, d_kc, nk, d_canch, mo->min_ksw_len, mo->bw, mo->fill_band_q4, mo->ext_max, mo->ext_band, mo->bw_long, d_nprob, (const int32_t*)nullptr, (DpProb*)nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemsetAsync(d_nprob + nk, 0, 4, st));
        TRY((dev_exclusive_scan<int32_t, int32_t>(ctx, d_nprob, d_poff, (size_t)nk + 1)));
        h_poff.resize(nk + 1);
        HIPCHK(hipMemcpyAsync(h_poff.data(), d_poff, (size_t)(nk + 1) * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        np = h_poff[nk];
        DpProb *d_probs;
        TRY(ctx_buf_t(ctx, "probs", (size_t)np, &d_probs));
        hipLaunchKernelGGL(k_segments_w<1>, dim3(nk), dim3(64), 0, st, d_kc, nk, d_canch, mo->min_ksw_len, mo->bw, mo->fill_band_q4, mo->ext_max, mo->ext_band, mo->bw_long, d_nprob, d_poff, d_probs);
        HIPCHK(hipGetLastError());
        t_sg.stop(); ht.mark("segments (sync: problems)");
        ctx->ctr.dp_problems += np;

        // ---- banded DP: narrow-band pass, then a wide-band pass for the problems whose path touched a band edge
        StageTimer t_dp(ctx, ST_DP, true);
        uint32_t *d_rawcig = nullptr; DpRes *d_res; int32_t *d_retry, *d_rcnt, *d_rlist;
        TRY(ctx_buf_t(ctx, "dp_res", (size_t)np, &d_res));
        TRY(ctx_buf_t(ctx, "retry_flag", (size_t)np + 1, &d_retry));
        TRY(ctx_buf_t(ctx, "retry_list", (size_t)np + 1, &d_rlist));
        TRY(ctx_buf_t(ctx, "retry_cnt", 4, &d_rcnt));
        HIPCHK(hipMemsetAsync(d_retry, 0, ((size_t)np + 1) * 4, st));
        HIPCHK(hipMemsetAsync(d_rcnt, 0, 16, st));
        TRY(dp_pass(ctx, qs, tg, mo, d_probs, np, d_res, &d_rawcig, d_retry, "", true));
        hipLaunchKernelGGL(k_retry_collect, dim3((np + 255) / 256), dim3(256), 0, st, d_retry, np, d_rcnt, d_rlist);
        HIPCHK(hipGetLastError());
        int32_t n_retry = 0;
        HIPCHK(hipMemcpyAsync(&n_retry, d_rcnt, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        { float ms = 0;
          if (hipEventElapsedTime(&ms, ctx->evk[5], ctx->evk[0]) == hipSuccess) ctx->stage_ms[ST_K_PK] += ms;
          if (hipEventElapsedTime(&ms, ctx->evk[0], ctx->evk[1]) == hipSuccess) ctx->stage_ms[ST_K_REG] += ms;
          if (hipEventElapsedTime(&ms, ctx->evk[3], ctx->evk[4]) == hipSuccess) ctx->stage_ms[ST_K_TRACEBACK] += ms; }
        if (n_retry > 0) {
            DpProb *d_probs2; DpRes *d_res2;
            TRY(ctx_buf_t(ctx, "probs_r", (size_t)n_retry, &d_probs2));
            TRY(ctx_buf_t(ctx, "dp_res_r", (size_t)n_retry, &d_res2));
            hipLaunchKernelGGL(k_retry_build, dim3((n_retry + 255) / 256), dim3(256), 0, st, d_probs, d_rlist, n_retry, mo->bw, mo->fill_band_q4, d_probs2);
            HIPCHK(hipGetLastError());
            TRY(dp_pass(ctx, qs, tg, mo, d_probs2, n_retry, d_res2, &d_rawcig, nullptr, "_r", false));
            hipLaunchKernelGGL(k_retry_merge, dim3((n_retry + 255) / 256), dim3(256), 0, st, d_probs2, d_res2, n_retry, d_res);
            HIPCHK(hipGetLastError());
        }
        ctx->dp_retries += n_retry;
        t_dp.stop(); ht.mark("dp passes");

        // ---- CIGARs of ALL kept chains are stitched now and travel to the (pinned) result buffer while the host does
        //      its second selection pass on the per-problem results; chains that pass drops leave unused gaps behind
        StageTimer t_g(ctx, ST_GATHER, true);
        {
            StitchRec *h_sv, *d_sv; int64_t *d_nfin, *d_foff; StitchProb *d_sp;
            TRY(ctx_hbuf_t(ctx, "h_stitch", (size_t)nk, &h_sv));
            TRY(ctx_buf_t(ctx, "stitch", (size_t)nk, &d_sv));
            TRY(ctx_buf_t(ctx, "stitch_n", (size_t)nk + 1, &d_nfin));
            TRY(ctx_buf_t(ctx, "stitch_off", (size_t)nk + 1, &d_foff));
            TRY(ctx_buf_t(ctx, "stitch_prob", (size_t)np, &d_sp));
            parallel_ranges(NT, nk, [&](int, int xa, int xb) {
                for (int x = xa; x < xb; ++x) {
                    const KeptLite &c = hl[x];
                    h_sv[x].p0 = h_poff[x]; h_sv[x].p1 = h_poff[x + 1]; h_sv[x].has_left = (c.qs > 0 && c.rs > 0) ? 1 : 0; h_sv[x].pad = 0;
                }
            });
            HIPCHK(hipMemcpyAsync(d_sv, h_sv, (size_t)nk * sizeof(StitchRec), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemsetAsync(d_sp, 0xff, (size_t)np * sizeof(StitchProb), st));
            hipLaunchKernelGGL(k_stitch_count, dim3(nk), dim3(64), 0, st, d_sv, nk, d_probs, d_res, d_rawcig, d_nfin, d_sp);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemsetAsync(d_nfin + nk, 0, 8, st));
            TRY((dev_exclusive_scan<int64_t, int64_t>(ctx, d_nfin, d_foff, (size_t)nk + 1)));
            TRY(ctx_hbuf_t(ctx, "h_stitch_off", (size_t)nk + 1, &h_foff));
            HIPCHK(hipMemcpyAsync(h_foff, d_foff, (size_t)(nk + 1) * 8, hipMemcpyDeviceToHost, st));
            ChainStat *d_cs; unsigned long long *d_acc;
            TRY(ctx_buf_t(ctx, "chain_stat", (size_t)nk, &d_cs));
            TRY(ctx_buf_t(ctx, "dp_acc", (size_t)TELR_N_DPCLS * 4 + 1, &d_acc));
            TRY(ctx_hbuf_t(ctx, "h_chain_stat", (size_t)nk, &h_cs));
            TRY(ctx_hbuf_t(ctx, "h_dp_acc", (size_t)TELR_N_DPCLS * 4 + 1, &h_acc));
            HIPCHK(hipMemsetAsync(d_acc, 0, ((size_t)TELR_N_DPCLS * 4 + 1) * 8, st));
            hipLaunchKernelGGL(k_chain_stats, dim3(nk), dim3(64), 0, st, d_sv, nk, d_res, d_cs, mo->cx_scale > 0 ? mo->cx_scale : 0);
            hipLaunchKernelGGL(k_dp_account, dim3((np + 255) / 256), dim3(256), 0, st, d_probs, d_res, np, d_acc);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(h_cs, d_cs, (size_t)nk * sizeof(ChainStat), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(h_acc, d_acc, ((size_t)TELR_N_DPCLS * 4 + 1) * 8, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            t_g.stop(); ht.mark("stitch count (sync)");                           // what follows is not waited for here
            const int64_t tot = h_foff[nk];
            uint32_t *d_fin;
            TRY(ctx_buf_t(ctx, "stitched", (size_t)tot + 1, &d_fin));
            // the stitched scratch is overwritten here: the previous call's DMA out of it must be over (it is, tens of ms ago)
            if (ctx->dma_inflight) HIPCHK(hipStreamWaitEvent(st, ctx->ev_dma, 0));
            hipLaunchKernelGGL(k_stitch_write, dim3((np + 31) / 32), dim3(256), 0, st, np, d_sp, d_probs, d_res, d_sv, d_rawcig, d_foff, d_fin);
            HIPCHK(hipGetLastError());
            if (!gate_enter(R, gate)) { ctx->err = "an earlier range of the call failed"; return TELR_E_HIP; }
            result_wait(R);                        // an earlier batch of this call may still be writing into the buffer that grows below
            cig_base = R->ncig;
            if (cig_base + (size_t)tot + 1 > R->cap) {
                if (!R->cig) pool_get(ctx, &R->cig, &R->cap);
                if (!cig_grow(&R->cig, &R->cap, cig_base, cig_base + (size_t)tot + 1 + (size_t)tot / 8)) return TELR_E_NOMEM;
            }
            R->ncig = cig_base + (size_t)tot;
            if (R->twin_n > cig_base) R->twin_n = cig_base;          // a range that was rolled back and runs again
            twin_put(ctx->twin_owner ? ctx->twin_owner : ctx, R, mo, cig_base, (size_t)tot, d_fin, ctx->twin_bases, st);
            if (tot) {
                // DMA on its own stream: the caller gets the records back while the CIGAR array is still travelling
                HIPCHK(hipEventRecord(ctx->ev_stitched, st));
                HIPCHK(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_stitched, 0));
                HIPCHK(hipMemcpyAsync(R->cig + cig_base, d_fin, (size_t)tot * 4, hipMemcpyDeviceToHost, ctx->copy_stream));
                HIPCHK(hipEventRecord(ctx->ev_dma, ctx->copy_stream)); ctx->dma_inflight = true;
                HIPCHK(hipEventCreateWithFlags(&R->dma_done, hipEventDisableTiming));
                HIPCHK(hipEventRecord(R->dma_done, ctx->copy_stream));
            }
        }

        // ---- host: per-chain numbers from the per-problem results (no op walking) ---------------------
        StageTimer t_as(ctx, ST_ASSEMBLE, false);
        for (int z = 0; z < TELR_N_DPCLS * 4; ++z) ctx->dpcls[z] += (int64_t)h_acc[z];
        for (int c = 0; c < TELR_N_DPCLS; ++c) ctx->ctr.dp_cells += (int64_t)h_acc[c * 4 + 1];
        ctx->ctr.window_bases += (int64_t)h_acc[TELR_N_DPCLS * 4];
        parallel_ranges(NT, nk, [&](int, int xa, int xb) {
            for (int x = xa; x < xb; ++x) {
                const KeptLite &c = hl[x]; telr_aln &r = kal[x]; const ChainStat &S = h_cs[x];
                const int qlen = r.qlen, tlen = r.tlen;
                int32_t qs_ = c.qs, rs_ = c.rs, qe_ = c.qe, re_ = c.re;
                const bool has_left = c.qs > 0 && c.rs > 0, has_right = c.qe < qlen && c.re < tlen;
                if (has_left) { qs_ = c.qs - S.l_bi; rs_ = c.rs - S.l_bj; }
                if (has_right) { qe_ = c.qe + S.r_bi; re_ = c.re + S.r_bj; }
                r.ts = rs_; r.te = re_;
                if (c.rev) { r.qs = qlen - qe_; r.qe = qlen - qs_; } else { r.qs = qs_; r.qe = qe_; }
                r.mlen = S.mlen; r.blen = S.blen; r.dp_score = S.dp;
            }
        });
        t_as.stop();
    }

    // ---- host: pass-2 selection, flags, mapq (threads over queries) --------------------------------------
    StageTimer t_as2(ctx, ST_ASSEMBLE, false);
    const bool per_t = (mo->flags & TELR_MF_PER_TARGET) != 0;
    // survivors of query q go to stage[k0 .. k0 + n_surv[q]) in rank order (a query keeps at most as many records as it had
    // kept chains), then one prefix sum over the queries places them in the result: no per-thread vectors, no re-copying
    if (ctx->h_stage.size() < (size_t)nk) ctx->h_stage.resize((size_t)nk + (size_t)nk / 8);
    if (ctx->h_nsurv.size() < (size_t)nq + 1) ctx->h_nsurv.resize((size_t)nq + 1);
    telr_aln *stage = ctx->h_stage.data(); int32_t *nsurv = ctx->h_nsurv.data();
    parallel_ranges(NT, nq, [&](int, int qa, int qb) {
        std::vector<Sel> s2; std::vector<int32_t> cscore, newidx, seen_tid;
        for (int q = qa; q < qb; ++q) {
            const int k0 = q_k0[q], n1 = q_k0[q + 1] - k0;
            s2.clear(); cscore.assign(n1, 0);
            for (int i = 0; i < n1; ++i) {
                const telr_aln &r = kal[k0 + i];
                cscore[i] = r.score;
                if ((mo->flags & TELR_MF_CIGAR) && r.dp_score < mo->min_dp_max) continue;
                Sel s; s.ci = i; s.key = r.dp_score; s.ord = (int32_t)s2.size(); s.tid = r.tid; s.fs = r.qs; s.fe = r.qe;
                s.parent = 0; s.subsc = 0; s.n_sub = 0; s.keep = 0;
                s2.push_back(s);
            }
            std::sort(s2.begin(), s2.end(), sel_less);
            select_chains(s2, mo, cscore);
            const int n2 = (int)s2.size();
            newidx.assign(n2, -1);
            bool seen_primary = false; seen_tid.clear();
            int nkp = 0;
            for (int i = 0; i < n2; ++i) if (s2[i].keep) newidx[i] = nkp++;
            int w = 0;
            for (int i = 0; i < n2; ++i) {
                if (!s2[i].keep) continue;
                const int x = k0 + s2[i].ci;
                telr_aln &r = stage[k0 + w++]; r = kal[x];
                r.parent = newidx[s2[i].parent]; r.subsc = s2[i].subsc; r.n_sub = s2[i].n_sub;
                if (s2[i].parent == i) {
                    bool first = true;
                    if (!per_t) { first = !seen_primary; seen_primary = true; }
                    else {
                        for (int32_t t2 : seen_tid) if (t2 == s2[i].tid) { first = false; break; }
                        if (first) seen_tid.push_back(s2[i].tid);
                    }
                    r.flags |= first ? TELR_F_PRIMARY : TELR_F_SUPPL;
                } else r.flags |= TELR_F_SECONDARY;
                r.mapq = mapq_of(r, mo);
                if (do_dp) { r.cigar_off = (int64_t)cig_base + h_foff[x]; r.n_cigar = (int32_t)(h_foff[x + 1] - h_foff[x]); }
            }
            nsurv[q] = w;
        }
    });
    std::vector<int64_t> out0((size_t)nq + 1);
    if (!gate_enter(R, gate)) { ctx->err = "an earlier range of the call failed"; return TELR_E_HIP; }
    const size_t r_base = R->alns.size();
    out0[0] = 0;
    for (int q = 0; q < nq; ++q) out0[q + 1] = out0[q] + nsurv[q];
    const int ns = (int)out0[nq];
    R->alns.resize(r_base + (size_t)ns);
    telr_aln *dst = R->alns.data() + r_base;
    std::vector<int64_t> tops(NT + 1, 0);
    parallel_ranges(NT, nq, [&](int t, int qa, int qb) {
        int64_t ops = 0;
        for (int q = qa; q < qb; ++q) {
            const int k0 = q_k0[q];
            for (int i = 0; i < nsurv[q]; ++i) { dst[out0[q] + i] = stage[k0 + i]; ops += stage[k0 + i].n_cigar; }
        }
        tops[t] = ops;
    });
    t_as2.stop();
    if (do_dp) {
        for (int t = 0; t < NT; ++t) ctx->ctr.cigar_ops += tops[t];
        // (the CIGAR DMA is waited for by whoever reads the CIGARs: result_wait)
    }
    ctx->ctr.records += ns;
    ht.mark("host assembly");
    stage_collect(ctx);
    ht.mark("stage_collect");
    return TELR_OK;
}

// One range [q0,q1) of the query set (at most one batch worth of bases); its records and CIGARs are appended to R.
// A range whose anchors do not fit int32 offsets (map_batch: TELR_SPLIT_RANGE) is halved and retried.
// (Rounds 1-3 could also map the longest reads of a range as a batch of their own on a worker context, and whole sub-batches on
// several: both lost to the walker-free back-tracking + two ranges in flight and were removed in round 4 -- docs/DESIGN_r1_r3.md.)
static int map_range(telr_ctx *ctx, const telr_index *ix, const telr_seqset *queries, const int32_t *qtarget, const int32_t *d_qt,
                     int32_t q0, int32_t q1, const telr_map_opt *mo, OccCut mid_occ, telr_result *R, int turn = -1)
{
    const int nq = q1 - q0;
    if (nq <= 0) return TELR_OK;
    int64_t total_bases = 0;
    for (int i = q0; i < q1; ++i) total_bases += queries->len[i];
    RangeTurn gate; gate.turn = turn;         // (pipelined ranges: R is only touched once the earlier ranges are in)
    int rr = map_batch(ctx, ix, queries, d_qt, q0, q1, mo, mid_occ, R, &gate);
    if (rr == TELR_OK) ctx->ctr.query_bases += total_bases;
    if (rr == TELR_SPLIT_RANGE) {
        if (nq < 2) { ctx->err = "one read seeds 2^31 anchors or more"; return TELR_E_RANGE; }
        int32_t mid = q0; int64_t acc = 0;
        while (mid < q1 - 1 && acc + queries->len[mid] <= total_bases / 2) acc += queries->len[mid++];
        if (mid == q0) mid = q0 + 1;
        TRY(map_range(ctx, ix, queries, qtarget, d_qt, q0, mid, mo, mid_occ, R, turn));
        return map_range(ctx, ix, queries, qtarget, d_qt, mid, q1, mo, mid_occ, R, turn);
    }
    return rr;
}

extern "C" int telr_map(telr_ctx *ctx, const telr_index *ix, const telr_seqset *queries, const int32_t *qtarget, const telr_map_opt *mo, telr_result **out)
{
    (void)hipGetLastError();          // a failed allocation of an EARLIER call leaves its error with the thread: not this call's
    if (!ctx || !ix || !queries || !mo || !out) return TELR_E_ARG;
    if (mo->chain_lookback != 64 && mo->chain_lookback != 128 && mo->chain_lookback != 256) { ctx->err = "chain_lookback must be 64, 128 or 256"; return TELR_E_ARG; }
    if (mo->e < mo->e2 || mo->q > mo->q2) { ctx->err = "two-piece gap cost needs e >= e2 and q <= q2"; return TELR_E_ARG; }
    if (mo->fill_margin < 0 || mo->fill_margin > 64) { ctx->err = "fill_margin must be 0..64"; return TELR_E_ARG; }
    if (mo->cx_scale < 0 || (mo->cx_scale > 0 && (mo->bw_long > 0 || mo->cx_scale > 64 || mo->cx_open < 0 || mo->cx_ext_min < 0 || mo->cx_ext_max < mo->cx_ext_min || mo->cx_decay < 0))) {
        ctx->err = "convex gap cost: cx_scale 1..64, 0 <= cx_ext_min <= cx_ext_max, cx_open >= 0, cx_decay >= 0, and no long join (bw_long) with it"; return TELR_E_ARG;
    }
    if (mo->vote_len != 0 && (mo->vote_len < 16 || mo->vote_len > 65536 || mo->vote_bin_shift < 0 || mo->vote_bin_shift > 20 || mo->vote_min < 1 || mo->vote_frac_q8 < 0 || mo->vote_frac_q8 > 256)) {
        ctx->err = "sub-read voting needs vote_len 16..65536, vote_bin_shift 0..20, vote_min >= 1, vote_frac_q8 0..256"; return TELR_E_ARG; }
    if (mo->flags & TELR_MF_FAITHFUL) { ctx->err = "TELR_MF_FAITHFUL is a mode of the CPU oracle (test infrastructure), not of the engine"; return TELR_E_ARG; }
    if (mo->max_gap >= TELR_TPAD || mo->ext_band * 2 + 1 > DP_DMAX || mo->ext_band < 1 || mo->ext_max < 1) return TELR_E_ARG;
    if (mo->bw_long > mo->bw && mo->ext_band > 31) {       // the two halves of a long-gap fill (spec 3.11) are bands of the extension width, held in 64 diagonals of LDS
        ctx->err = "long join (bw_long) needs ext_band <= 31"; return TELR_E_ARG; }
    if (queries->max_len >= (1 << 24)) return TELR_E_RANGE;
    HIPCHK(hipSetDevice(ctx->device));
    memset(ctx->stage_ms, 0, sizeof(ctx->stage_ms));
    memset(&ctx->ctr, 0, sizeof(ctx->ctr));
    memset(ctx->dpcls, 0, sizeof(ctx->dpcls));
    ctx->dp_retries = 0; ctx->pk_launches = 0; ctx->st_pending = 0;
    const int nq = queries->n;
    if (qtarget) for (int i = 0; i < nq; ++i) if (qtarget[i] >= ix->targets->n) return TELR_E_ARG;
    int32_t *d_qt = nullptr;
    if (qtarget && nq > 0) {
        TRY(ctx_buf_t(ctx, "qtarget", (size_t)nq, &d_qt));
        HIPCHK(hipMemcpyAsync(d_qt, qtarget, (size_t)nq * 4, hipMemcpyHostToDevice, ctx->stream)); HIPCHK(hipStreamSynchronize(ctx->stream));      // not the null stream: it waits for all blocking streams of the process
    }
    OccCut mid_occ; mid_occ.mid_occ = index_mid_occ(ix, mo); mid_occ.d_tmid = nullptr;
    if (qtarget || (mo->flags & TELR_MF_PER_TARGET)) TRY(index_per_target_occ(ctx, ix, mo, &mid_occ.d_tmid));
    telr_result *R = new telr_result();
    R->ctx = ctx;
    if (!ctx->aln_pool.empty()) {                 // the largest recycled record array
        size_t best = 0;
        for (size_t i = 1; i < ctx->aln_pool.size(); ++i) if (ctx->aln_pool[i].capacity() > ctx->aln_pool[best].capacity()) best = i;
        R->alns = std::move(ctx->aln_pool[best]); R->alns.clear();
        ctx->aln_pool.erase(ctx->aln_pool.begin() + best);
    }
    const auto t_wall0 = std::chrono::steady_clock::now();
    int64_t total_bases = 0;
    for (int i = 0; i < nq; ++i) total_bases += queries->len[i];
    ctx->twin_owner = ctx; ctx->twin_bases = total_bases;
    {
        // Ranges bounded by bases: a read set of any size streams through as consecutive ranges.  HBM is 288 GB and a range
        // needs ~75 B of scratch per read base at 0.25 anchors per base: a read set of up to 1.6 Gbp is ONE range (configs[2]
        // reads alone in ranges of 0.5 / 1 / 1.4 / 2.1 Gbp: 13.7 / 14.9 / 15.3 / 15.5 Gbp/s -- fewer synchronisation points and
        // tails; 151 GB in use at 2.1), a larger one is cut into ranges of at most 1.4 Gbp that run two at a time (below; the
        // scratch of context and second slot is grow-only: ~115 + ~100 GB at this density).  What really bounds a range is its
        // anchors (int32 offsets, ~50 B each): ranges hold at most 1.6 G anchors (0.8 G each when two are in flight) at the
        // density the last call on the index has seen -- before any call, at an upper bound computed from the index's
        // occurrence counts; a range that overflows all the same is halved by map_range.
        int64_t batch_bases = 1600LL << 20;
        bool fixed = false;
        if (const char *e = getenv("TELR_BATCH_MBP")) { long v = atol(e); if (v > 0) { batch_bases = (int64_t)v << 20; fixed = true; } }
        if (const char *e = getenv("TELR_BATCH_KBP")) { long v = atol(e); if (v > 0) { batch_bases = (int64_t)v << 10; fixed = true; } }     // tests
        // Range pipelining: a read set that needs more than one range runs TWO ranges at a time on two slots (the context and
        // a second one of the same kind), so the host work between the stages of a range -- synchronisations, the second
        // selection pass, the record assembly -- and its latency-bound stretches are covered by the other range's kernels;
        // results are appended in range order through the turn gate of the result (round 2: configs[2] from 15.7 to 16.5-17.4 Gbp/s,
        // ranges of 0.7-1.6 Gbp: flat).  A read set that fits ONE range is halved when it holds 0.67 Gbp or more (below that the halves lose: 5-17 % at
        // 0.4-0.5 Gbp).  TELR_PIPELINE=1 switches it off; =force pipelines any multi-range call (tests).
        int pipe = 2; bool force = false;
        if (const char *e = getenv("TELR_PIPELINE")) { force = !strcmp(e, "force"); pipe = force || atoi(e) >= 2 ? 2 : 1; }
        // Round 6: a call with per-query targets (S6: every window read against the forward and the reverse-complement contig of its locus,
        // 0.75-0.9 Gbp; the polishing map) runs its ranges one at a time: it shares the device with the other calls of the loci pass already,
        // and on the hard genome its ranges are a few long chaining / sorting kernels on reads that bring 10^6 anchors each -- two in flight took
        // 0.96 or 1.45 s per 1,000 c2r loci from pass to pass, one at a time 0.89; configs[2]: 96 -> 90 ms (profiles/r06_chain_loop_choice_ab.txt, part 8).
        if (ctx->pipe_nomem || (!force && (ctx->debug || nq < 4000))) pipe = 1;
        const bool in_turn = pipe == 2 && qtarget && !force;          // one range, or the ranges of the plan below in turn (no more scratch per range than two in flight took)
        if (pipe == 2 && !fixed) {
            // ranges of at most 1.4 Gbp (two in flight: ~200 GB of scratch at configs[2]'s anchor density) and at most 1.6 G
            // anchors at the density seen by the last call on this index; a read set within one such range is not split
            int64_t cap = 1400LL << 20;
            // sub-read voting carries ~25 B per query base more (hits staged at 8 B each, compacted minimizers): two 1.4-Gbp
            // ranges in flight fill the device (2 x 152 GB measured at configs[3]) and leave the BAM writer nothing
            if (mo->vote_len > 0 && !qtarget && !(mo->flags & TELR_MF_PER_TARGET)) cap = 1100LL << 20;
            const double per_base = ix->anchors_per_base > 0 ? ix->anchors_per_base : index_density_bound(ix, mid_occ.mid_occ);
            if (per_base > 0) cap = std::min<int64_t>(cap, std::max<int64_t>(256LL << 20, (int64_t)(0.8e9 / per_base)));      // two in flight: half the anchor budget each
            // ... or that is large enough for two halves in flight to win: measured on configs[2] reads, two ranges against one:
            // 0.40 Gbp 30.7 / 27.4 ms, 0.51 Gbp 32.9 / 33.9, 0.81 Gbp 47.2 / 51.3, 1.01 Gbp 57.3 / 63.2 (the shard of a 4-rank run)
            if (total_bases > std::min<int64_t>(batch_bases, (int64_t)(per_base > 0 ? 1.6e9 / per_base : 1e18)) || total_bases >= (640LL << 20) || force) {
                int64_t nr = std::max<int64_t>(2, (total_bases + cap - 1) / cap);
                nr += nr & 1;          // an even number of equal ranges keeps both slots busy to the end (configs[2]: 3 ranges 215 ms, 4 ranges 205 ms per step)
                batch_bases = (total_bases + nr - 1) / nr + queries->max_len + 1;      // the slack keeps the greedy cut below from leaving a stub range behind
            } else pipe = 1;
        }
        std::vector<std::pair<int32_t, int32_t>> ranges;
        for (int32_t q0 = 0; q0 < nq; ) {
            int32_t q1 = q0; int64_t b = 0;
            while (q1 < nq && (q1 == q0 || b + queries->len[q1] <= batch_bases)) { b += queries->len[q1]; ++q1; }
            ranges.push_back(std::make_pair(q0, q1));
            q0 = q1;
        }
        { static const bool tr = trace_on("host");
          if (tr) fprintf(stderr, "[host plan] %d queries, %.1f Mbp, %zu range(s) of <= %.1f Mbp, %s, anchors per base seen %.3f%s\n", nq, total_bases / 1048576.0, ranges.size(), batch_bases / 1048576.0,
                          pipe == 2 && ranges.size() >= 2 && !in_turn ? "two in flight" : "one at a time", ix->anchors_per_base, qtarget ? ", per-query targets" : ""); }
        if (in_turn && ranges.size() >= 2) {
            // ONE range when the device has room for it (what such a call waits for is its longest chaining run, once per range: two ranges
            // in turn took 1.57 s per 1,000 c2r loci, one range 0.88), the ranges of the plan above in turn when it has not
            int r = total_bases <= (1600LL << 20) ? map_range(ctx, ix, queries, qtarget, d_qt, 0, nq, mo, mid_occ, R) : TELR_E_NOMEM;
            if (r == TELR_OK && getenv("TELR_TEST_PIPE_NOMEM")) r = TELR_E_NOMEM;      // tests: exercise the fall-back
            if (r == TELR_E_NOMEM) {
                (void)hipGetLastError();          // (the failed allocation's error is sticky for this thread)
                mem_note(ctx, "telr_map: one range with per-query targets ran out");
                (void)hipDeviceSynchronize();
                result_wait(R); R->alns.clear(); R->ncig = 0; R->twin_n = 0;
                { std::lock_guard<std::mutex> lk(R->gate_m); R->turn = 0; }
                memset(ctx->stage_ms, 0, sizeof(ctx->stage_ms)); memset(&ctx->ctr, 0, sizeof(ctx->ctr)); memset(ctx->dpcls, 0, sizeof(ctx->dpcls));
                ctx->dp_retries = 0; ctx->pk_launches = 0; ctx->st_pending = 0; ctx->err.clear();
                r = TELR_OK;
                for (size_t i = 0; i < ranges.size() && r == TELR_OK; ++i) r = map_range(ctx, ix, queries, qtarget, d_qt, ranges[i].first, ranges[i].second, mo, mid_occ, R);
            }
            if (r != TELR_OK) { delete R; return r; }
            if (ctx->ctr.anchors > 0 && ctx->ctr.query_bases > (64LL << 20)) ix->anchors_per_base = (double)ctx->ctr.anchors / (double)ctx->ctr.query_bases;
        } else if (pipe == 2 && ranges.size() >= 2) {
            if (!ctx->slot1) {
                int r = ctx_init(ctx->device, ctx->background, &ctx->slot1);
                if (r != TELR_OK) { delete R; return r; }
            }
            telr_ctx *P[2] = { ctx, ctx->slot1 };
            P[1]->twin_owner = ctx; P[1]->twin_bases = total_bases;
            { telr_ctx *c = P[1]; memset(c->stage_ms, 0, sizeof(c->stage_ms)); memset(&c->ctr, 0, sizeof(c->ctr)); memset(c->dpcls, 0, sizeof(c->dpcls)); c->dp_retries = 0; c->pk_launches = 0; c->st_pending = 0; c->err.clear(); }
            int rc[2] = { TELR_OK, TELR_OK };
            std::atomic<size_t> next_range{0};
            auto slot = [&](int s) {
                (void)hipSetDevice(ctx->device);
                for (size_t i; (i = next_range.fetch_add(1)) < ranges.size(); ) {        // whichever slot is free takes the next range; results are appended in range order (turn gate)
                    int r = map_range(P[s], ix, queries, qtarget, d_qt, ranges[i].first, ranges[i].second, mo, mid_occ, R, (int)i);
                    gate_leave(R, (int)i, r == TELR_OK);
                    if (r != TELR_OK) { rc[s] = r; return; }
                }
            };
            std::thread t1(slot, 1);
            slot(0);
            t1.join();
            // the failing range's error, not that of the range it made leave
            int prc = TELR_OK;
            for (int s = 0; s < 2 && prc == TELR_OK; ++s) if (rc[s] != TELR_OK && P[s]->err != "an earlier range of the call failed") { if (s) ctx->err = P[1]->err; prc = rc[s]; }
            for (int s = 0; s < 2 && prc == TELR_OK; ++s) if (rc[s] != TELR_OK) { if (s) ctx->err = P[1]->err; prc = rc[s]; }
            const bool test_nomem = prc == TELR_OK && getenv("TELR_TEST_PIPE_NOMEM");      // tests: exercise the fall-back below
            if (test_nomem) prc = TELR_E_NOMEM;
            if (prc == TELR_E_NOMEM) {
                if (!test_nomem) ctx->pipe_nomem = true;
                // two ranges in flight did not fit (a device shared with something else, a denser index than the hint said, or
                // the BAM writer's buffers of an earlier call still held): give the second slot's scratch and the writer's
                // buffers back and run the call again one range at a time, in ranges of 1 Gbp at most
                (void)hipGetLastError();          // the failed allocation's error is sticky for this thread: the next launch check would report it again
                mem_note(ctx, "telr_map: two ranges in flight ran out");
                telr_destroy(ctx->slot1); ctx->slot1 = nullptr;
                (void)hipDeviceSynchronize();
                for (auto &kv : ctx->bufs) if (kv.first.compare(0, 4, "bam_") == 0 && kv.second.p) { (void)hipFree(kv.second.p); kv.second.p = nullptr; kv.second.bytes = 0; }
                result_wait(R); R->alns.clear(); R->ncig = 0; R->twin_n = 0;
                { std::lock_guard<std::mutex> lk(R->gate_m); R->turn = 0; }
                mem_note(ctx, "telr_map: after giving back slot 2 + writer");
                memset(ctx->stage_ms, 0, sizeof(ctx->stage_ms)); memset(&ctx->ctr, 0, sizeof(ctx->ctr)); memset(ctx->dpcls, 0, sizeof(ctx->dpcls));
                ctx->dp_retries = 0; ctx->pk_launches = 0; ctx->st_pending = 0;
                int64_t lim = std::min<int64_t>(batch_bases, 1024LL << 20);
                prc = TELR_OK;
                for (int32_t q0 = 0; q0 < nq && prc == TELR_OK; ) {
                    int32_t q1 = q0; int64_t b = 0;
                    while (q1 < nq && (q1 == q0 || b + queries->len[q1] <= lim)) { b += queries->len[q1]; ++q1; }
                    prc = map_range(ctx, ix, queries, qtarget, d_qt, q0, q1, mo, mid_occ, R);
                    q0 = q1;
                }
                if (prc != TELR_OK) { delete R; return prc; }
            } else if (prc != TELR_OK) { delete R; return prc; }
            else
            { telr_ctx *c = P[1];
              for (int z = 0; z < TELR_N_STAGES; ++z) ctx->stage_ms[z] += c->stage_ms[z];
              const int64_t *src = (const int64_t*)&c->ctr; int64_t *dst = (int64_t*)&ctx->ctr;
              for (size_t z = 0; z < sizeof(telr_counters) / 8; ++z) dst[z] += src[z];
              for (int z = 0; z < TELR_N_DPCLS * 4; ++z) ctx->dpcls[z] += c->dpcls[z];
              ctx->dp_retries += c->dp_retries; ctx->pk_launches += c->pk_launches; }
            if (ctx->ctr.anchors > 0 && ctx->ctr.query_bases > (64LL << 20)) ix->anchors_per_base = (double)ctx->ctr.anchors / (double)ctx->ctr.query_bases;
        } else {
            auto limit_for = [&](double per_base) { return std::min<int64_t>(batch_bases, std::max<int64_t>(256LL << 20, (int64_t)(1.6e9 / per_base))); };
            int64_t limit = batch_bases;
            if (!fixed) { const double pb = ix->anchors_per_base > 0 ? ix->anchors_per_base : index_density_bound(ix, mid_occ.mid_occ); if (pb > 0) limit = limit_for(pb); }
            for (int32_t q0 = 0; q0 < nq; ) {
                int32_t q1 = q0; int64_t b = 0;
                while (q1 < nq && (q1 == q0 || b + queries->len[q1] <= limit)) { b += queries->len[q1]; ++q1; }
                int r = map_range(ctx, ix, queries, qtarget, d_qt, q0, q1, mo, mid_occ, R);
                if (r != TELR_OK) { delete R; return r; }
                if (ctx->ctr.anchors > 0 && ctx->ctr.query_bases > (64LL << 20)) {
                    ix->anchors_per_base = (double)ctx->ctr.anchors / (double)ctx->ctr.query_bases;
                    if (!fixed) limit = limit_for(ix->anchors_per_base);
                }
                q0 = q1;
            }
        }
    }
    ctx->stage_ms[ST_MAP_WALL] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_wall0).count();
    *out = R;
    return TELR_OK;
}

// ---------------------------------------------------------------------------------------
// Window reads (the `read_type="all"` selection of prep_assembly_inputs, TELR_assembly.py:384-415): for every window
// (target id, [lo, hi)) the ascending, distinct query ids with ANY record -- primary, secondary or supplementary --
// that overlaps it (ts < hi and te > lo).  The reference runs one pysam fetch per locus on the sorted stage-1 BAM; here
// the records are still in memory (any telr_aln array: a result's, or records gathered from several).  Host code: one
// pass over the records by the worker pool, windows sorted by (target, lo) with a running maximum of hi so that a
// record stops scanning at the first window that cannot reach it.
extern "C" int telr_window_reads(const telr_aln *recs, int64_t n_rec, int32_t n_win, const int32_t *win_tid, const int32_t *win_lo, const int32_t *win_hi,
                                 int64_t *out_off, int32_t *out_qid, int64_t cap, int64_t *needed)
{
    if (n_rec < 0 || n_win < 0 || (n_rec > 0 && !recs) || (n_win > 0 && (!win_tid || !win_lo || !win_hi)) || !out_off || !needed || cap < 0 || (cap > 0 && !out_qid)) return TELR_E_ARG;
    if (n_rec >= (1LL << 31)) return TELR_E_RANGE;
    std::vector<int32_t> ord(n_win);
    for (int i = 0; i < n_win; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { const int32_t la = win_lo[a] > 0 ? win_lo[a] : 0, lb = win_lo[b] > 0 ? win_lo[b] : 0;
        return win_tid[a] != win_tid[b] ? win_tid[a] < win_tid[b] : la != lb ? la < lb : a < b; });
    std::vector<int64_t> key(n_win); std::vector<int32_t> hmax(n_win), hi(n_win);
    for (int j = 0; j < n_win; ++j) {
        const int w = ord[j];
        key[j] = (int64_t)win_tid[w] << 32 | (uint32_t)(win_lo[w] > 0 ? win_lo[w] : 0); hi[j] = win_hi[w];
        hmax[j] = (j > 0 && win_tid[ord[j - 1]] == win_tid[w] && hmax[j - 1] > hi[j]) ? hmax[j - 1] : hi[j];      // per target
    }
    const int NT = host_threads();
    std::vector<std::vector<uint64_t>> part(NT);           // (sorted window << 32 | qid), in record order inside a part
    parallel_ranges(NT, (int)n_rec, [&](int t, int a, int b) {
        std::vector<uint64_t> &P = part[t];
        for (int x = a; x < b; ++x) {
            const telr_aln &r = recs[x];
            if (r.tid < 0) continue;
            // windows of the record's target starting before its end: [.., j1)
            const int64_t k = (int64_t)r.tid << 32 | (uint32_t)r.te;
            int j1 = (int)(std::lower_bound(key.begin(), key.end(), k) - key.begin());
            for (int j = j1 - 1; j >= 0 && (key[j] >> 32) == r.tid && hmax[j] > r.ts; --j)
                if (hi[j] > r.ts) P.push_back((uint64_t)j << 32 | (uint32_t)r.qid);
        }
    });
    // per window: count, place (parts in record order), then sort + unique the (short) lists
    std::vector<int64_t> cnt((size_t)n_win + 1, 0);
    for (int t = 0; t < NT; ++t) for (uint64_t v : part[t]) ++cnt[(size_t)(v >> 32) + 1];
    for (int j = 0; j < n_win; ++j) cnt[j + 1] += cnt[j];
    std::vector<int32_t> all((size_t)cnt[n_win]);
    { std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
      for (int t = 0; t < NT; ++t) for (uint64_t v : part[t]) all[(size_t)fill[v >> 32]++] = (int32_t)(uint32_t)v; }
    std::vector<int64_t> uniq(n_win, 0);
    parallel_ranges(NT, n_win, [&](int, int a, int b) {
        for (int j = a; j < b; ++j) {
            int32_t *f = all.data() + cnt[j], *l = all.data() + cnt[j + 1];
            std::sort(f, l);
            uniq[j] = std::unique(f, l) - f;
        }
    });
    std::vector<int64_t> len_of(n_win, 0), src_of(n_win, 0);
    for (int j = 0; j < n_win; ++j) { len_of[ord[j]] = uniq[j]; src_of[ord[j]] = cnt[j]; }
    out_off[0] = 0;
    for (int w = 0; w < n_win; ++w) out_off[w + 1] = out_off[w] + len_of[w];
    *needed = out_off[n_win];
    if (*needed > cap) return TELR_E_RANGE;             // the caller retries with *needed
    for (int w = 0; w < n_win; ++w) if (len_of[w]) memcpy(out_qid + out_off[w], all.data() + src_of[w], (size_t)len_of[w] * 4);
    return TELR_OK;
}

// ---------------------------------------------------------------------------------------
// debug taps for the stage-level parity tests (last batch of the last telr_map call)
extern "C" int64_t telr_debug_dp_retries(const telr_ctx *ctx) { return ctx ? ctx->dp_retries : 0; }
extern "C" int64_t telr_debug_pk_launches(const telr_ctx *ctx) { return ctx ? ctx->pk_launches : 0; }
extern "C" int64_t telr_debug_n_anchor(const telr_ctx *ctx) { return ctx ? ctx->dbg_na : 0; }
extern "C" int telr_debug_fetch(telr_ctx *ctx, const char *what, void *dst, int64_t bytes)
{
    if (!ctx || !what || !dst) return TELR_E_ARG;
    auto it = ctx->bufs.find(what);
    if (it == ctx->bufs.end() || !it->second.p || (size_t)bytes > it->second.bytes) return TELR_E_ARG;
    HIPCHK(hipMemcpy(dst, it->second.p, (size_t)bytes, hipMemcpyDeviceToHost));
    return TELR_OK;
}
extern "C" int64_t telr_debug_n_chain(const telr_ctx *ctx) { return ctx ? (int64_t)ctx->dbg_chain.size() / 9 : 0; }
extern "C" const int32_t *telr_debug_chains(const telr_ctx *ctx) { return ctx ? ctx->dbg_chain.data() : nullptr; }
extern "C" int telr_debug_index(telr_ctx *ctx, const telr_index *ix, uint64_t *ent_hash, uint32_t *ent_off, uint32_t *pos)
{
    if (!ctx || !ix) return TELR_E_ARG;
    if (ent_hash) HIPCHK(hipMemcpy(ent_hash, ix->d_ent_hash, (size_t)ix->n_ent * 8, hipMemcpyDeviceToHost));
    if (ent_off) HIPCHK(hipMemcpy(ent_off, ix->d_ent_off, ((size_t)ix->n_ent + 1) * 4, hipMemcpyDeviceToHost));
    if (pos) HIPCHK(hipMemcpy(pos, ix->d_pos, (size_t)ix->n_mz * 4, hipMemcpyDeviceToHost));
    return TELR_OK;
}
extern "C" int32_t telr_debug_mid_occ(const telr_index *ix, const telr_map_opt *mo) { return ix && mo ? index_mid_occ(ix, mo) : -1; }

// ---------------------------------------------------------------------------------------
// depth medians
extern "C" int telr_depth_medians(telr_ctx *ctx, const telr_result *r, int32_t n_targets, const int32_t *target_len, int32_t n_iv,
                                  const int32_t *iv_tid, const int32_t *iv_start, const int32_t *iv_end, double *median_out)
{
    (void)hipGetLastError();          // a failed allocation of an EARLIER call leaves its error with the thread: not this call's
    if (!ctx || !r || n_targets <= 0 || !target_len || n_iv < 0 || (n_iv && (!iv_tid || !iv_start || !iv_end || !median_out))) return TELR_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<int64_t> toff(n_targets + 1, 0);
    for (int i = 0; i < n_targets; ++i) toff[i + 1] = toff[i] + target_len[i] + 1;
    for (int i = 0; i < n_iv; ++i) if (iv_tid[i] < 0 || iv_tid[i] >= n_targets) return TELR_E_ARG;
    std::vector<DepthRec> recs;
    for (const telr_aln &a : r->alns) {
        if (a.flags & TELR_F_SECONDARY) continue;
        if (a.tid < 0 || a.tid >= n_targets) return TELR_E_ARG;
        DepthRec d; d.tid = a.tid; d.ts = a.ts; d.n_cigar = a.n_cigar; d.pad = 0; d.cigar_off = a.cigar_off; recs.push_back(d);
    }
    int32_t *d_diff, *d_tlen, *d_ivt, *d_ivs, *d_ive; int64_t *d_toff; DepthRec *d_recs; uint32_t *d_cig; double *d_out;
    TRY(ctx_buf_t(ctx, "dm_diff", (size_t)toff[n_targets] + 1, &d_diff));
    TRY(ctx_buf_t(ctx, "dm_toff", (size_t)n_targets + 1, &d_toff));
    TRY(ctx_buf_t(ctx, "dm_tlen", (size_t)n_targets, &d_tlen));
    TRY(ctx_buf_t(ctx, "dm_recs", recs.size(), &d_recs));
    result_wait(r);
    // the CIGAR array: the result's own device copy when it kept one (TELR_MF_KEEP_CIGARS), else uploaded
    const bool twin = r->d_cig && !r->twin_off && r->twin_n == r->ncig;
    if (twin) { d_cig = r->d_cig; HIPCHK(hipDeviceSynchronize()); }      // (its last pieces were copied on other streams, as in telr_write_bam_dev)
    else TRY(ctx_buf_t(ctx, "dm_cig", r->ncig, &d_cig));
    TRY(ctx_buf_t(ctx, "dm_ivt", (size_t)n_iv, &d_ivt));
    TRY(ctx_buf_t(ctx, "dm_ivs", (size_t)n_iv, &d_ivs));
    TRY(ctx_buf_t(ctx, "dm_ive", (size_t)n_iv, &d_ive));
    TRY(ctx_buf_t(ctx, "dm_out", (size_t)n_iv, &d_out));
    HIPCHK(hipMemsetAsync(d_diff, 0, ((size_t)toff[n_targets] + 1) * 4, st));
    HIPCHK(hipMemcpyAsync(d_toff, toff.data(), (size_t)(n_targets + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_tlen, target_len, (size_t)n_targets * 4, hipMemcpyHostToDevice, st));
    if (!recs.empty()) HIPCHK(hipMemcpyAsync(d_recs, recs.data(), recs.size() * sizeof(DepthRec), hipMemcpyHostToDevice, st));
    if (r->ncig && !twin) HIPCHK(hipMemcpyAsync(d_cig, r->cig, r->ncig * 4, hipMemcpyHostToDevice, st));
    if (n_iv) {
        HIPCHK(hipMemcpyAsync(d_ivt, iv_tid, (size_t)n_iv * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_ivs, iv_start, (size_t)n_iv * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_ive, iv_end, (size_t)n_iv * 4, hipMemcpyHostToDevice, st));
    }
    if (!recs.empty()) hipLaunchKernelGGL(k_depth_diff, dim3(((int)recs.size() + 63) / 64), dim3(64), 0, st, d_recs, (int32_t)recs.size(), d_cig, d_toff, d_diff);
    TRY(dev_inclusive_scan_i32(ctx, d_diff, (size_t)toff[n_targets]));
    if (n_iv) {
        hipLaunchKernelGGL(k_depth_median, dim3(n_iv), dim3(256), 0, st, d_diff, d_toff, d_tlen, n_iv, d_ivt, d_ivs, d_ive, d_out);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(median_out, d_out, (size_t)n_iv * 8, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    return TELR_OK;
}

// ---------------------------------------------------------------------------------------
// PAF / SAM emitters (host).  The reference's call sites consume PAF columns 0,1,4,5,7,8,9,10,11
// (TELR_liftover.py:215-240,356-380; TELR_te.py:89-95,136-142) and SAM records with NM/MD/AS/SA/cs
// (hand-off H1: Sniffles, samtools depth, pysam; docs/02_Usage.md:76).
struct SeqView { const char *ascii; const int64_t *off; const int32_t *len; };
static const char COMP_TAB[256] = {
#define C16 'N','N','N','N','N','N','N','N','N','N','N','N','N','N','N','N'
    C16, C16, C16, C16,
    'N','T','N','G','N','N','N','C','N','N','N','N','N','N','N','N','N','N','N','N','A','A','N','N','N','N','N','N','N','N','N','N',
    'N','t','N','g','N','N','N','c','N','N','N','N','N','N','N','N','N','N','N','N','a','a','N','N','N','N','N','N','N','N','N','N',
    C16, C16, C16, C16, C16, C16, C16, C16
};
// bases as the engine sees them (the 2-bit packing: A C G T/U in either case, anything else ambiguous): what NM / MD / cs
// print, so that the text writers and the device-side BAM writer (bam_dev.hip.h) agree byte for byte
static inline char up(char c)
{
    switch (c) { case 'A': case 'a': return 'A'; case 'C': case 'c': return 'C'; case 'G': case 'g': return 'G'; case 'T': case 't': case 'U': case 'u': return 'T'; default: return 'N'; }
}

static void cigar_text(const uint32_t *cg, int n, int clip5, int clip3, char clipc, std::string &out)
{
    char buf[24];
    if (clip5 > 0) { snprintf(buf, sizeof(buf), "%d%c", clip5, clipc); out += buf; }
    for (int i = 0; i < n; ++i) { snprintf(buf, sizeof(buf), "%u%c", cg[i] >> 4, "MID"[cg[i] & 0xf]); out += buf; }
    if (clip3 > 0) { snprintf(buf, sizeof(buf), "%d%c", clip3, clipc); out += buf; }
}

extern "C" int telr_write_paf(const telr_result *r, const char *const *qnames, const char *const *tnames, int with_cigar,
                              const char *path, int append)
{
    if (!r || !qnames || !tnames) return TELR_E_ARG;
    result_wait(r);
    FILE *f = path ? fopen(path, append ? "a" : "w") : stdout;
    if (!f) return TELR_E_ARG;
    std::string line;
    for (const telr_aln &a : r->alns) {
        char buf[512];
        snprintf(buf, sizeof(buf), "%s\t%d\t%d\t%d\t%c\t%s\t%d\t%d\t%d\t%d\t%d\t%d\tNM:i:%d\tAS:i:%d\ttp:A:%c\tcm:i:%d\ts1:i:%d",
                 qnames[a.qid], a.qlen, a.qs, a.qe, (a.flags & TELR_F_REV) ? '-' : '+', tnames[a.tid], a.tlen, a.ts, a.te, a.mlen, a.blen, a.mapq,
                 a.blen - a.mlen, a.dp_score, (a.flags & TELR_F_SECONDARY) ? 'S' : 'P', a.cnt, a.score);
        line = buf;
        if (!(a.flags & TELR_F_SECONDARY)) { snprintf(buf, sizeof(buf), "\ts2:i:%d", a.subsc); line += buf; }
        if (with_cigar && a.n_cigar > 0) { line += "\tcg:Z:"; cigar_text(r->cig + a.cigar_off, a.n_cigar, 0, 0, 'S', line); }
        line += '\n';
        fwrite(line.data(), 1, line.size(), f);
    }
    if (path) fclose(f);
    return TELR_OK;
}

extern "C" int telr_write_sam(const telr_result *r, int32_t n_queries, const char *const *qnames, const char *q_ascii, const int64_t *q_off,
                              const int32_t *q_len, int32_t n_targets, const char *const *tnames, const char *t_ascii, const int64_t *t_off,
                              const int32_t *t_len, int32_t flags, const char *rg_id, const char *rg_sm, const char *rg_lb,
                              const char *pg_line, const char *path)
{
    if (!r || !qnames || !q_ascii || !q_off || !q_len || !tnames || !t_ascii || !t_off || !t_len) return TELR_E_ARG;
    result_wait(r);
    FILE *f = path ? fopen(path, "w") : stdout;
    if (!f) return TELR_E_ARG;
    const bool sorted = (flags & TELR_SAM_SORTED) != 0, prim_only = (flags & TELR_SAM_PRIMARY_ONLY) != 0;
    if (!(flags & TELR_SAM_NO_HEADER)) {
        fprintf(f, sorted ? "@HD\tVN:1.6\tSO:coordinate\n" : "@HD\tVN:1.6\tSO:unsorted\tGO:query\n");
        for (int t = 0; t < n_targets; ++t) fprintf(f, "@SQ\tSN:%s\tLN:%d\n", tnames[t], t_len[t]);
        if (rg_id) fprintf(f, "@RG\tID:%s\tSM:%s\tLB:%s\n", rg_id, rg_sm ? rg_sm : rg_id, rg_lb ? rg_lb : "lib");
        fprintf(f, "@PG\tID:telr_amd\tPN:telr_amd\tVN:0.1.0\tCL:%s\n", pg_line ? pg_line : "telr_map");
    }
    // TELR_SAM_SORTED: lines are collected with their (target, position) key and written in coordinate order, unmapped
    // reads last (what `samtools sort | samtools view` prints: refID, position, forward strand before reverse; stable, so
    // remaining ties keep the query order)
    std::vector<std::pair<int64_t, std::string>> keyed;
    auto emit = [&](int64_t key, const std::string &l) { if (sorted) keyed.emplace_back(key, l); else fwrite(l.data(), 1, l.size(), f); };
    // records are sorted by (qid, rank); group per query
    const size_t n = r->alns.size();
    size_t i = 0;
    std::string seq, rc, line, md, cs, sa;
    for (int q = 0; q < n_queries; ++q) {
        size_t j = i;
        while (j < n && r->alns[j].qid == q) ++j;
        const char *qs = q_ascii + q_off[q]; const int ql = q_len[q];
        if (j == i) {
            if (!(flags & TELR_SAM_NO_UNMAPPED)) {
                line.clear(); line += qnames[q]; line += "\t4\t*\t0\t0\t*\t*\t0\t0\t"; line.append(qs, (size_t)ql); line += "\t*";
                if (rg_id) { line += "\tRG:Z:"; line += rg_id; }
                line += '\n';
                emit(INT64_MAX, line);
            }
            continue;
        }
        rc.resize(ql);
        for (int x = 0; x < ql; ++x) rc[x] = COMP_TAB[(unsigned char)qs[ql - 1 - x]];
        for (size_t k = i; k < j; ++k) {
            const telr_aln &a = r->alns[k];
            const bool rev = (a.flags & TELR_F_REV) != 0, sec = (a.flags & TELR_F_SECONDARY) != 0, sup = (a.flags & TELR_F_SUPPL) != 0;
            if (prim_only && (sec || sup)) continue;                 // samtools view -F0x900
            const char *qstr = rev ? rc.data() : qs;                 // query on the alignment strand
            const int clip5 = rev ? ql - a.qe : a.qs, clip3 = rev ? a.qs : ql - a.qe;
            const uint32_t *cg = r->cig + a.cigar_off;
            const char *ts = t_ascii + t_off[a.tid];
            // NM / MD / cs from the CIGAR walk
            int nm = 0, qi = clip5, ti = a.ts, run = 0;
            md.clear(); cs.clear();
            char buf[32];
            for (int z = 0; z < a.n_cigar; ++z) {
                const int op = cg[z] & 0xf, l = (int)(cg[z] >> 4);
                if (op == 0) {
                    int csrun = 0;
                    for (int x = 0; x < l; ++x) {
                        const char qc = up(qstr[qi + x]), tc = up(ts[ti + x]);
                        if (qc == tc && tc != 'N') { ++run; ++csrun; }
                        else {
                            ++nm;
                            if (flags & TELR_SAM_MD) { snprintf(buf, sizeof(buf), "%d%c", run, tc); md += buf; }
                            run = 0;
                            if (flags & TELR_SAM_CS) { if (csrun) { snprintf(buf, sizeof(buf), ":%d", csrun); cs += buf; csrun = 0; } cs += '*'; cs += (char)(tc | 32); cs += (char)(qc | 32); }
                        }
                    }
                    if ((flags & TELR_SAM_CS) && csrun) { snprintf(buf, sizeof(buf), ":%d", csrun); cs += buf; }
                    qi += l; ti += l;
                } else if (op == 1) {
                    nm += l;
                    if (flags & TELR_SAM_CS) { cs += '+'; for (int x = 0; x < l; ++x) cs += (char)(up(qstr[qi + x]) | 32); }
                    qi += l;
                } else {
                    nm += l;
                    if (flags & TELR_SAM_MD) { snprintf(buf, sizeof(buf), "%d^", run); md += buf; for (int x = 0; x < l; ++x) md += up(ts[ti + x]); run = 0; }
                    if (flags & TELR_SAM_CS) { cs += '-'; for (int x = 0; x < l; ++x) cs += (char)(up(ts[ti + x]) | 32); }
                    ti += l;
                }
            }
            if (flags & TELR_SAM_MD) { snprintf(buf, sizeof(buf), "%d", run); md += buf; }
            const bool hard = sup && !(flags & TELR_SAM_SOFTCLIP);      // secondary: SEQ '*' and soft clips, as minimap2 without --secondary-seq
            int fl = (rev ? 0x10 : 0) | (sec ? 0x100 : 0) | (sup ? 0x800 : 0);
            line.clear();
            line += qnames[q];
            snprintf(buf, sizeof(buf), "\t%d\t", fl); line += buf;
            line += tnames[a.tid];
            snprintf(buf, sizeof(buf), "\t%d\t%d\t", a.ts + 1, a.mapq); line += buf;
            if (a.n_cigar > 0) cigar_text(cg, a.n_cigar, clip5, clip3, hard ? 'H' : 'S', line); else line += '*';
            line += "\t*\t0\t0\t";
            if (sec) line += '*';
            else if (hard) line.append(qstr + clip5, (size_t)(ql - clip5 - clip3));
            else line.append(qstr, (size_t)ql);
            line += "\t*";
            snprintf(buf, sizeof(buf), "\tNM:i:%d\tAS:i:%d", nm, a.dp_score); line += buf;
            if (flags & TELR_SAM_MD) { line += "\tMD:Z:"; line += md; }
            if (flags & TELR_SAM_CS) { line += "\tcs:Z:"; line += cs; }
            // SA: the other primary / supplementary records of this read
            if (!sec) {
                sa.clear();
                for (size_t k2 = i; k2 < j; ++k2) {
                    const telr_aln &b = r->alns[k2];
                    if (k2 == k || (b.flags & TELR_F_SECONDARY)) continue;
                    const bool brev = (b.flags & TELR_F_REV) != 0;
                    const int b5 = brev ? ql - b.qe : b.qs, b3 = brev ? b.qs : ql - b.qe;
                    int nI = 0, nD = 0;
                    for (int z = 0; z < b.n_cigar; ++z) { uint32_t c = r->cig[b.cigar_off + z]; if ((c & 0xf) == 1) nI += c >> 4; else if ((c & 0xf) == 2) nD += c >> 4; }
                    char sb[256];
                    const int mlen_q = (b.qe - b.qs);
                    snprintf(sb, sizeof(sb), "%s,%d,%c,", tnames[b.tid], b.ts + 1, brev ? '-' : '+'); sa += sb;
                    if (b5) { snprintf(sb, sizeof(sb), "%dS", b5); sa += sb; }
                    snprintf(sb, sizeof(sb), "%dM", mlen_q - nI); sa += sb;
                    if (nI) { snprintf(sb, sizeof(sb), "%dI", nI); sa += sb; }
                    if (nD) { snprintf(sb, sizeof(sb), "%dD", nD); sa += sb; }
                    if (b3) { snprintf(sb, sizeof(sb), "%dS", b3); sa += sb; }
                    snprintf(sb, sizeof(sb), ",%d,%d;", b.mapq, b.blen - b.mlen); sa += sb;
                }
                if (!sa.empty()) { line += "\tSA:Z:"; line += sa; }
            }
            snprintf(buf, sizeof(buf), "\ttp:A:%c\tcm:i:%d\ts1:i:%d", sec ? 'S' : 'P', a.cnt, a.score); line += buf;
            if (!sec) { snprintf(buf, sizeof(buf), "\ts2:i:%d", a.subsc); line += buf; }
            if (rg_id) { line += "\tRG:Z:"; line += rg_id; }
            line += '\n';
            emit(((int64_t)(a.tid + 1) << 33) | (int64_t)(uint32_t)a.ts << 1 | (rev ? 1 : 0), line);
        }
        i = j;
    }
    if (sorted) {
        std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<int64_t, std::string> &x, const std::pair<int64_t, std::string> &y) { return x.first < y.first; });
        for (auto &kv : keyed) fwrite(kv.second.data(), 1, kv.second.size(), f);
    }
    if (path) fclose(f);
    return TELR_OK;
}

// ---------------------------------------------------------------------------------------
// Coordinate-sorted BAM + BAI (replaces `samtools sort -o BAM SAM; samtools index BAM`,
// reference src/telr/TELR_alignment.py:103-114; hand-off H1 to Sniffles / pysam).
#include <zlib.h>
static inline int reg2bin(int64_t beg, int64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}
static inline void put32(std::string &s, uint32_t v) { s.append((const char*)&v, 4); }
static inline void put16(std::string &s, uint16_t v) { s.append((const char*)&v, 2); }
static inline uint8_t nt16(char c)
{
    switch (c) { case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'G': case 'g': return 4; case 'T': case 't': case 'U': case 'u': return 8; default: return 15; }
}
static bool bgzf_block(const char *src, size_t n, int level, std::string &out)
{
    uLong bound = compressBound((uLong)n) + 64;
    out.resize(18 + bound + 8);
    z_stream zs; memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    zs.next_in = (Bytef*)src; zs.avail_in = (uInt)n; zs.next_out = (Bytef*)&out[18]; zs.avail_out = (uInt)bound;
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); return false; }
    size_t clen = zs.total_out; deflateEnd(&zs);
    static const uint8_t hdr[12] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0 };
    memcpy(&out[0], hdr, 12);
    out[12] = 'B'; out[13] = 'C'; out[14] = 2; out[15] = 0;
    uint16_t bsize = (uint16_t)(clen + 25);
    memcpy(&out[16], &bsize, 2);
    uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)src, (uInt)n), isize = (uint32_t)n;
    memcpy(&out[18 + clen], &crc, 4); memcpy(&out[18 + clen + 4], &isize, 4);
    out.resize(18 + clen + 8);
    return true;
}

extern "C" int telr_write_bam(const telr_result *r, int32_t n_queries, const char *const *qnames, const char *q_ascii, const int64_t *q_off,
                              const int32_t *q_len, int32_t n_targets, const char *const *tnames, const char *t_ascii, const int64_t *t_off,
                              const int32_t *t_len, int32_t flags, const char *rg_id, const char *rg_sm, const char *rg_lb,
                              const char *pg_line, const char *bam_path, int32_t write_index, int32_t level)
{
    if (!r || !qnames || !q_ascii || !q_off || !q_len || !tnames || !t_ascii || !t_off || !t_len || !bam_path) return TELR_E_ARG;
    result_wait(r);
    const size_t n = r->alns.size();
    // --- 1. binary records (one per alignment + one per unmapped read), built in parallel over queries
    std::vector<size_t> qfirst((size_t)n_queries + 1, 0);
    { size_t i = 0; for (int q = 0; q < n_queries; ++q) { qfirst[q] = i; while (i < n && r->alns[i].qid == q) ++i; } qfirst[n_queries] = i; }
    std::vector<int> q_unmapped;
    for (int q = 0; q < n_queries; ++q) if (qfirst[q] == qfirst[q + 1] && !(flags & TELR_SAM_NO_UNMAPPED)) q_unmapped.push_back(q);
    const size_t nrec = n + q_unmapped.size();
    std::vector<std::string> recs(nrec);
    std::vector<int64_t> key(nrec);          // (refID+1)<<32 | pos ; unmapped last
    const int NT = host_threads();
    parallel_ranges(NT, n_queries, [&](int, int qa, int qb) {
        std::string rc, md, cs, sa, tagbuf;
        char buf[64];
        for (int q = qa; q < qb; ++q) {
            const size_t i0 = qfirst[q], i1 = qfirst[q + 1];
            if (i0 == i1) continue;
            const char *qs = q_ascii + q_off[q]; const int ql = q_len[q];
            rc.resize(ql);
            for (int x = 0; x < ql; ++x) rc[x] = COMP_TAB[(unsigned char)qs[ql - 1 - x]];
            for (size_t k = i0; k < i1; ++k) {
                const telr_aln &a = r->alns[k];
                const bool rev = (a.flags & TELR_F_REV) != 0, sec = (a.flags & TELR_F_SECONDARY) != 0, sup = (a.flags & TELR_F_SUPPL) != 0;
                const char *qstr = rev ? rc.data() : qs;
                const int clip5 = rev ? ql - a.qe : a.qs, clip3 = rev ? a.qs : ql - a.qe;
                const uint32_t *cg = r->cig + a.cigar_off;
                const char *ts = t_ascii + t_off[a.tid];
                int nm = 0, qi = clip5, ti = a.ts, run = 0;
                md.clear(); cs.clear();
                for (int z = 0; z < a.n_cigar; ++z) {
                    const int op = cg[z] & 0xf, l = (int)(cg[z] >> 4);
                    if (op == 0) {
                        int csrun = 0;
                        for (int x = 0; x < l; ++x) {
                            const char qc = up(qstr[qi + x]), tc = up(ts[ti + x]);
                            if (qc == tc && tc != 'N') { ++run; ++csrun; }
                            else {
                                ++nm;
                                if (flags & TELR_SAM_MD) { snprintf(buf, sizeof(buf), "%d%c", run, tc); md += buf; }
                                run = 0;
                                if (flags & TELR_SAM_CS) { if (csrun) { snprintf(buf, sizeof(buf), ":%d", csrun); cs += buf; csrun = 0; } cs += '*'; cs += (char)(tc | 32); cs += (char)(qc | 32); }
                            }
                        }
                        if ((flags & TELR_SAM_CS) && csrun) { snprintf(buf, sizeof(buf), ":%d", csrun); cs += buf; }
                        qi += l; ti += l;
                    } else if (op == 1) {
                        nm += l;
                        if (flags & TELR_SAM_CS) { cs += '+'; for (int x = 0; x < l; ++x) cs += (char)(up(qstr[qi + x]) | 32); }
                        qi += l;
                    } else {
                        nm += l;
                        if (flags & TELR_SAM_MD) { snprintf(buf, sizeof(buf), "%d^", run); md += buf; for (int x = 0; x < l; ++x) md += up(ts[ti + x]); run = 0; }
                        if (flags & TELR_SAM_CS) { cs += '-'; for (int x = 0; x < l; ++x) cs += (char)(up(ts[ti + x]) | 32); }
                        ti += l;
                    }
                }
                if (flags & TELR_SAM_MD) { snprintf(buf, sizeof(buf), "%d", run); md += buf; }
                const bool hard = sup && !(flags & TELR_SAM_SOFTCLIP);      // secondary: SEQ '*' and soft clips, as minimap2 without --secondary-seq
                const int fl = (rev ? 0x10 : 0) | (sec ? 0x100 : 0) | (sup ? 0x800 : 0);
                const int seq_lo = sec ? 0 : (hard ? clip5 : 0), seq_hi = sec ? 0 : (hard ? ql - clip3 : ql), l_seq = seq_hi - seq_lo;
                std::vector<uint32_t> bc;
                if (clip5 > 0) bc.push_back((uint32_t)clip5 << 4 | (hard ? 5u : 4u));
                for (int z = 0; z < a.n_cigar; ++z) bc.push_back(cg[z]);          // M=0 I=1 D=2 as in BAM
                if (clip3 > 0) bc.push_back((uint32_t)clip3 << 4 | (hard ? 5u : 4u));
                const bool long_cigar = bc.size() > 65535;
                tagbuf.clear();
                auto tag_i = [&](const char *t, int32_t v) { tagbuf += t; tagbuf += 'i'; tagbuf.append((const char*)&v, 4); };
                auto tag_z = [&](const char *t, const std::string &v) { tagbuf += t; tagbuf += 'Z'; tagbuf += v; tagbuf += '\0'; };
                tag_i("NM", nm); tag_i("AS", a.dp_score);
                if (flags & TELR_SAM_MD) tag_z("MD", md);
                if (flags & TELR_SAM_CS) tag_z("cs", cs);
                if (!sec) {
                    sa.clear();
                    for (size_t k2 = i0; k2 < i1; ++k2) {
                        const telr_aln &b = r->alns[k2];
                        if (k2 == k || (b.flags & TELR_F_SECONDARY)) continue;
                        const bool brev = (b.flags & TELR_F_REV) != 0;
                        const int b5 = brev ? ql - b.qe : b.qs, b3 = brev ? b.qs : ql - b.qe;
                        int nI = 0, nD = 0;
                        for (int z = 0; z < b.n_cigar; ++z) { uint32_t c = r->cig[b.cigar_off + z]; if ((c & 0xf) == 1) nI += c >> 4; else if ((c & 0xf) == 2) nD += c >> 4; }
                        char sb[256];
                        snprintf(sb, sizeof(sb), "%s,%d,%c,", tnames[b.tid], b.ts + 1, brev ? '-' : '+'); sa += sb;
                        if (b5) { snprintf(sb, sizeof(sb), "%dS", b5); sa += sb; }
                        snprintf(sb, sizeof(sb), "%dM", (b.qe - b.qs) - nI); sa += sb;
                        if (nI) { snprintf(sb, sizeof(sb), "%dI", nI); sa += sb; }
                        if (nD) { snprintf(sb, sizeof(sb), "%dD", nD); sa += sb; }
                        if (b3) { snprintf(sb, sizeof(sb), "%dS", b3); sa += sb; }
                        snprintf(sb, sizeof(sb), ",%d,%d;", b.mapq, b.blen - b.mlen); sa += sb;
                    }
                    if (!sa.empty()) tag_z("SA", sa);
                }
                tagbuf += "tpA"; tagbuf += sec ? 'S' : 'P';
                tag_i("cm", a.cnt); tag_i("s1", a.score);
                if (!sec) tag_i("s2", a.subsc);
                if (rg_id) tag_z("RG", rg_id);
                if (long_cigar) {       // -L: real CIGAR in a CG:B,I tag, placeholder <l_seq>S<ref_len>N in the record
                    tagbuf += "CGBI"; uint32_t cnt = (uint32_t)bc.size(); tagbuf.append((const char*)&cnt, 4); tagbuf.append((const char*)bc.data(), bc.size() * 4);
                }
                std::string &o = recs[k];
                const size_t l_name = strlen(qnames[q]) + 1;
                const uint32_t n_cig = long_cigar ? 2u : (uint32_t)bc.size();
                o.reserve(36 + l_name + n_cig * 4 + (l_seq + 1) / 2 + l_seq + tagbuf.size());
                put32(o, 0);                                         // block_size, patched below
                put32(o, (uint32_t)a.tid); put32(o, (uint32_t)a.ts);
                o += (char)(uint8_t)l_name; o += (char)(uint8_t)a.mapq; put16(o, (uint16_t)reg2bin(a.ts, a.te > a.ts ? a.te : a.ts + 1));
                put16(o, (uint16_t)n_cig); put16(o, (uint16_t)fl);
                put32(o, (uint32_t)l_seq); put32(o, (uint32_t)-1); put32(o, (uint32_t)-1); put32(o, 0);
                o.append(qnames[q], l_name);
                if (long_cigar) { put32(o, (uint32_t)l_seq << 4 | 4u); put32(o, (uint32_t)(a.te - a.ts) << 4 | 3u); }
                else o.append((const char*)bc.data(), bc.size() * 4);
                for (int x = 0; x < l_seq; x += 2) { uint8_t hi = nt16(qstr[seq_lo + x]), lo2 = x + 1 < l_seq ? nt16(qstr[seq_lo + x + 1]) : 0; o += (char)(hi << 4 | lo2); }
                o.append((size_t)l_seq, (char)0xff);
                o += tagbuf;
                uint32_t bs = (uint32_t)o.size() - 4; memcpy(&o[0], &bs, 4);
                key[k] = ((int64_t)(a.tid + 1) << 33) | (int64_t)(uint32_t)a.ts << 1 | (rev ? 1 : 0);
            }
        }
    });
    for (size_t u = 0; u < q_unmapped.size(); ++u) {
        const int q = q_unmapped[u]; const char *qs = q_ascii + q_off[q]; const int ql = q_len[q];
        std::string &o = recs[n + u];
        const size_t l_name = strlen(qnames[q]) + 1;
        put32(o, 0); put32(o, (uint32_t)-1); put32(o, (uint32_t)-1);
        o += (char)(uint8_t)l_name; o += (char)0; put16(o, 4680); put16(o, 0); put16(o, 4);
        put32(o, (uint32_t)ql); put32(o, (uint32_t)-1); put32(o, (uint32_t)-1); put32(o, 0);
        o.append(qnames[q], l_name);
        for (int x = 0; x < ql; x += 2) { uint8_t hi = nt16(qs[x]), lo2 = x + 1 < ql ? nt16(qs[x + 1]) : 0; o += (char)(hi << 4 | lo2); }
        o.append((size_t)ql, (char)0xff);
        if (rg_id) { o += "RGZ"; o += rg_id; o += '\0'; }
        uint32_t bs = (uint32_t)o.size() - 4; memcpy(&o[0], &bs, 4);
        key[n + u] = INT64_MAX;
    }
    // --- 2. coordinate sort (stable)
    std::vector<uint32_t> order(nrec);
    for (size_t i = 0; i < nrec; ++i) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return key[x] < key[y]; });
    // --- 3. uncompressed stream: header + records, cut into BGZF blocks of <= 65280 bytes
    std::string head;
    {
        std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
        char b[512];
        for (int t = 0; t < n_targets; ++t) { snprintf(b, sizeof(b), "@SQ\tSN:%s\tLN:%d\n", tnames[t], t_len[t]); text += b; }
        if (rg_id) { snprintf(b, sizeof(b), "@RG\tID:%s\tSM:%s\tLB:%s\n", rg_id, rg_sm ? rg_sm : rg_id, rg_lb ? rg_lb : "lib"); text += b; }
        text += "@PG\tID:telr_amd\tPN:telr_amd\tVN:0.1.0\tCL:"; text += pg_line ? pg_line : "telr_map"; text += "\n";
        head += "BAM\1"; put32(head, (uint32_t)text.size()); head += text; put32(head, (uint32_t)n_targets);
        for (int t = 0; t < n_targets; ++t) { uint32_t ln = (uint32_t)strlen(tnames[t]) + 1; put32(head, ln); head.append(tnames[t], ln); put32(head, (uint32_t)t_len[t]); }
    }
    const size_t BLK = 65280;
    std::vector<uint64_t> ustart(nrec + 1);        // uncompressed offset of every record (sorted order)
    uint64_t upos = head.size();
    for (size_t i = 0; i < nrec; ++i) { ustart[i] = upos; upos += recs[order[i]].size(); }
    ustart[nrec] = upos;
    const uint64_t utotal = upos;
    std::string ubuf; ubuf.resize(utotal);
    memcpy(&ubuf[0], head.data(), head.size());
    parallel_ranges(NT, (int)nrec, [&](int, int a0, int a1) { for (int i = a0; i < a1; ++i) memcpy(&ubuf[ustart[i]], recs[order[i]].data(), recs[order[i]].size()); });
    const size_t nblk = (size_t)((utotal + BLK - 1) / BLK);
    std::vector<std::string> cblk(nblk);
    bool ok = true;
    parallel_ranges(NT, (int)nblk, [&](int, int b0, int b1) {
        for (int b = b0; b < b1; ++b) { size_t o = (size_t)b * BLK, l = std::min(BLK, (size_t)utotal - o); if (!bgzf_block(&ubuf[o], l, level > 0 ? level : 1, cblk[b])) ok = false; }
    });
    if (!ok) return TELR_E_NOMEM;
    std::vector<uint64_t> coff(nblk + 1, 0);
    for (size_t b = 0; b < nblk; ++b) coff[b + 1] = coff[b] + cblk[b].size();
    FILE *f = fopen(bam_path, "wb");
    if (!f) return TELR_E_ARG;
    for (size_t b = 0; b < nblk; ++b) fwrite(cblk[b].data(), 1, cblk[b].size(), f);
    static const uint8_t eof_blk[28] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    fwrite(eof_blk, 1, 28, f);
    fclose(f);
    if (!write_index) return TELR_OK;
    // --- 4. BAI
    auto voff = [&](uint64_t u) { size_t b = (size_t)(u / BLK); if (b >= nblk) return (uint64_t)(coff[nblk] << 16); return (uint64_t)(coff[b] << 16 | (u - (uint64_t)b * BLK)); };
    std::string bai = "BAI\1"; put32(bai, (uint32_t)n_targets);
    size_t i = 0;
    uint64_t n_no_coor = q_unmapped.size();
    for (int t = 0; t < n_targets; ++t) {
        std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
        const int n_lin = (t_len[t] >> 14) + 1;
        std::vector<uint64_t> lin(n_lin, 0);
        int max_lin = 0;
        uint64_t ref_beg = 0, ref_end = 0, n_mapped = 0;
        bool any = false;
        while (i < nrec && key[order[i]] != INT64_MAX && (int)((key[order[i]] >> 33) - 1) == t) {
            const telr_aln &a = r->alns[order[i]];
            const uint64_t vb = voff(ustart[i]), ve = voff(ustart[i + 1]);
            const uint32_t bin = (uint32_t)reg2bin(a.ts, a.te > a.ts ? a.te : a.ts + 1);
            auto &ch = bins[bin];
            if (!ch.empty() && (ch.back().second >> 16) == (vb >> 16)) ch.back().second = ve; else ch.push_back(std::make_pair(vb, ve));
            const int w0 = a.ts >> 14, w1 = ((a.te > a.ts ? a.te : a.ts + 1) - 1) >> 14;
            for (int wv = w0; wv <= w1 && wv < n_lin; ++wv) { if (lin[wv] == 0 || vb < lin[wv]) lin[wv] = vb; if (wv + 1 > max_lin) max_lin = wv + 1; }
            if (!any) { ref_beg = vb; any = true; }
            ref_end = ve; ++n_mapped; ++i;
        }
        put32(bai, (uint32_t)(bins.size() + (any ? 1 : 0)));
        for (auto &kv : bins) {
            put32(bai, kv.first); put32(bai, (uint32_t)kv.second.size());
            for (auto &c : kv.second) { bai.append((const char*)&c.first, 8); bai.append((const char*)&c.second, 8); }
        }
        if (any) {   // samtools' metadata pseudo-bin 37450
            put32(bai, 37450u); put32(bai, 2u);
            bai.append((const char*)&ref_beg, 8); bai.append((const char*)&ref_end, 8);
            uint64_t zero = 0; bai.append((const char*)&n_mapped, 8); bai.append((const char*)&zero, 8);
        }
        // linear index: fill empty windows with the next lower filled value (as samtools does)
        for (int wv = 1; wv < max_lin; ++wv) if (lin[wv] == 0) lin[wv] = lin[wv - 1];
        put32(bai, (uint32_t)max_lin);
        for (int wv = 0; wv < max_lin; ++wv) bai.append((const char*)&lin[wv], 8);
    }
    bai.append((const char*)&n_no_coor, 8);
    std::string bai_path = std::string(bam_path) + ".bai";
    f = fopen(bai_path.c_str(), "wb");
    if (!f) return TELR_E_ARG;
    fwrite(bai.data(), 1, bai.size(), f);
    fclose(f);
    return TELR_OK;
}

#include "bam_dev.hip.h"
#include "fasta_io.hip.h"
#include "pileup.hip.h"
#include "poa.hip.h"
