// pileup.hip.h — pile-up consensus of the targets from the primary records of a result (spec 3.12; SURVEY 8(f) rank 4: the
// polishing hand-off H3, `samtools view -F0x900 sorted.bam | wtpoa-cns -d CNS -i -`, src/telr/TELR_assembly.py:226-247).
// A majority vote per target position -- NOT wtpoa-cns's partial-order alignment: it is offered behind a flag of
// telr_assembly (polish="pileup") because a different consensus algorithm can change call sets.
//   k_pile_count  one wave per primary record: walks the CIGAR 64 ops per trip against the 2-bit reads and adds to the cell
//                 of every target position it covers: base votes of M columns, D columns, the first CONS_KMAX inserted bases
//                 right after the position (global atomics; ~40 reads deep, so contention is mild)
//   k_pile_call   one thread per target position: how many bases the position emits (0 dropped, 1, + inserted columns)
//   rocPRIM scan, k_pile_write: the consensus strings
// Included at the end of telr_engine.hip (after bam_dev.hip.h, whose strand-aware 32-base fetch it uses).
#pragma once
#define CONS_KMAX 8
#define CONS_MAXDEL 30     /* a longer D is a structural difference (a read of the other allele), not an error of the draft: no vote */
#define CONS_CELL (6 + CONS_KMAX + 4 * CONS_KMAX)       /* u32 per position: b[4], del, nq, insn[K], insb[K][4] */

__global__ void __launch_bounds__(64) k_pile_count(const telr_aln *__restrict__ alns, int32_t n, const uint32_t *__restrict__ cig,
                                                   const uint32_t *__restrict__ q2, const uint32_t *__restrict__ qn, const int64_t *__restrict__ qboff,
                                                   const int64_t *__restrict__ tbase, uint32_t *__restrict__ cell)
{
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= n) return;
    const telr_aln a = alns[k];
    if (a.tid < 0 || (a.flags & (TELR_F_SECONDARY | TELR_F_SUPPL))) return;
    const int rev = (a.flags & TELR_F_REV) ? 1 : 0, ql = a.qlen;
    const int64_t qb0 = qboff[a.qid], tb = tbase[a.tid];
    const uint32_t *__restrict__ cg = cig + a.cigar_off;
    int qi0 = rev ? ql - a.qe : a.qs, ti0 = a.ts;
    for (int z0 = 0; z0 < a.n_cigar; z0 += 64) {
        const int z = z0 + lane;
        const bool have = z < a.n_cigar;
        const uint32_t c = have ? cg[z] : 0u;
        const int op = (int)(c & 0xfu), L = (int)(c >> 4);
        const int qadv = (have && op != 2) ? L : 0, tadv = (have && op != 1) ? L : 0;
        const int qinc = d_wave_incl(qadv, lane), tinc = d_wave_incl(tadv, lane);
        const int qi = qi0 + qinc - qadv, ti = ti0 + tinc - tadv;
        if (have) {
            if (op == 0) {
                for (int p0 = 0; p0 < L; p0 += 32) {
                    const B32 qw = d_strand32(q2, qn, qb0, ql, rev, qi + p0);
                    const int m = L - p0 < 32 ? L - p0 : 32;
                    uint32_t *cc = cell + (tb + ti + p0) * CONS_CELL;
                    for (int j = 0; j < m; ++j) atomicAdd(cc + (int64_t)j * CONS_CELL + ((qw.n >> j & 1u) ? 5 : (int)((qw.w >> (2 * j)) & 3u)), 1u);
                }
            } else if (op == 2) {
                uint32_t *cc = cell + (tb + ti) * CONS_CELL + 4;
                if (L <= CONS_MAXDEL) for (int x = 0; x < L; ++x) atomicAdd(cc + (int64_t)x * CONS_CELL, 1u);
            } else if (ti > a.ts) {
                const B32 qw = d_strand32(q2, qn, qb0, ql, rev, qi);
                uint32_t *cc = cell + (tb + ti - 1) * CONS_CELL;
                const int m = L < CONS_KMAX ? L : CONS_KMAX;
                for (int x = 0; x < m; ++x) {
                    atomicAdd(cc + 6 + x, 1u);
                    if (!(qw.n >> x & 1u)) atomicAdd(cc + 6 + CONS_KMAX + 4 * x + (int)((qw.w >> (2 * x)) & 3u), 1u);
                }
            }
        }
        qi0 += __shfl(qinc, 63); ti0 += __shfl(tinc, 63);
    }
}
// bases emitted by global target position g (targets laid end to end without padding); draft code 0..3 or 4
__device__ __forceinline__ int d_pile_call(const uint32_t *__restrict__ cc, int draft, int min_depth, uint8_t *out)
{
    const uint32_t cov = cc[0] + cc[1] + cc[2] + cc[3] + cc[4] + cc[5];
    int n = 0;
    if ((int)cov < min_depth) { if (out) out[0] = (uint8_t)"ACGTN"[draft]; return 1; }
    if (2u * cc[4] <= cov) {
        int best = -1; uint32_t bv = 0;
        for (int b = 0; b < 4; ++b) if (cc[b] > bv) { bv = cc[b]; best = b; }
        int ch = draft;
        if (best >= 0) { if (draft < 4 && cc[draft] == bv) best = draft; ch = best; }
        if (out) out[n] = (uint8_t)"ACGTN"[ch];
        ++n;
    }
    for (int k = 0; k < CONS_KMAX && 2u * cc[6 + k] > cov; ++k) {
        int best = -1; uint32_t bv = 0;
        for (int b = 0; b < 4; ++b) { const uint32_t v = cc[6 + CONS_KMAX + 4 * k + b]; if (v > bv) { bv = v; best = b; } }
        if (out) out[n] = best >= 0 ? (uint8_t)"ACGT"[best] : (uint8_t)'N';
        ++n;
    }
    return n;
}
// target of a global position (tbase ascending, tbase[nt] = total)
__device__ __forceinline__ int d_pile_target(const int64_t *__restrict__ tbase, int nt, int64_t g)
{
    int lo = 0, hi = nt - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tbase[mid] <= g) lo = mid; else hi = mid - 1; }
    return lo;
}
template <int WRITE>
__global__ void __launch_bounds__(256) k_pile_call(const uint32_t *__restrict__ cell, int64_t total, const int64_t *__restrict__ tbase, int32_t nt,
                                                   const uint32_t *__restrict__ t2, const uint32_t *__restrict__ tn, const int64_t *__restrict__ tboff,
                                                   int32_t min_depth, int32_t *__restrict__ emit, const int64_t *__restrict__ woff, uint8_t *__restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) { if (!WRITE && g == total) emit[g] = 0; return; }
    const int t = d_pile_target(tbase, nt, g);
    const int draft = d_base(t2, tn, tboff[t] + (g - tbase[t]));
    if (!WRITE) emit[g] = d_pile_call(cell + g * CONS_CELL, draft, min_depth, nullptr);
    else { uint8_t tmp[1 + CONS_KMAX]; const int n = d_pile_call(cell + g * CONS_CELL, draft, min_depth, tmp); uint8_t *o = out + woff[g]; for (int i = 0; i < n; ++i) o[i] = tmp[i]; }
}
__global__ void __launch_bounds__(256) k_widen_i32(const int32_t *__restrict__ in, int64_t n, int64_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}
// the scan at the first position of every target (and at the end): the targets' offsets in the output, for ONE copy back
__global__ void __launch_bounds__(256) k_pick_i64(const int64_t *__restrict__ in, const int64_t *__restrict__ at, int32_t n, int64_t *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[at[i]];
}

struct telr_consensus { std::string seq; std::vector<int64_t> off; std::vector<int32_t> len; };
extern "C" void telr_consensus_free(telr_consensus *c) { delete c; }
extern "C" int32_t telr_consensus_count(const telr_consensus *c) { return c ? (int32_t)c->len.size() : 0; }
extern "C" const char *telr_consensus_seq(const telr_consensus *c) { return c ? c->seq.data() : nullptr; }
extern "C" const int64_t *telr_consensus_off(const telr_consensus *c) { return c ? c->off.data() : nullptr; }
extern "C" const int32_t *telr_consensus_len(const telr_consensus *c) { return c ? c->len.data() : nullptr; }

static int consensus_impl(telr_ctx *ctx, const telr_result *r, const telr_seqset *queries, const telr_index *idx, int32_t min_depth, telr_consensus **out);
extern "C" int telr_consensus_build(telr_ctx *ctx, const telr_result *r, const telr_seqset *queries, const telr_index *idx, int32_t min_depth, telr_consensus **out)
{
    (void)hipGetLastError();          // a failed allocation of an EARLIER call leaves its error with the thread: not this call's
    int rc = consensus_impl(ctx, r, queries, idx, min_depth, out);
    if (rc == TELR_E_NOMEM) {
        // the device is full of the grow-only mapping scratch of earlier calls (configs[3]): it goes back, largest buffers first
        // (the pile-up cells are 184 bytes per contig base), and the call runs once more
        (void)hipGetLastError();
        uint64_t need = 0;
        if (idx && idx->targets) need = (uint64_t)idx->targets->total_bases * CONS_CELL * 4 + ((uint64_t)256 << 20);
        mem_note(ctx, "telr_consensus_build: out of memory");
        ctx_release_map_scratch(ctx, need);
        rc = consensus_impl(ctx, r, queries, idx, min_depth, out);
    }
    return rc;
}
static int consensus_impl(telr_ctx *ctx, const telr_result *r, const telr_seqset *queries, const telr_index *idx, int32_t min_depth, telr_consensus **out)
{
    if (!ctx || !r || !queries || !idx || !idx->targets || !out || min_depth < 0) return TELR_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    result_wait(r);
    hipStream_t st = ctx->stream;
    const telr_seqset *tg = idx->targets;
    const int32_t nt = tg->n;
    const size_t n = r->alns.size();
    for (const telr_aln &a : r->alns) if (a.qid < 0 || a.qid >= queries->n || a.tid < 0 || a.tid >= nt) return TELR_E_ARG;
    std::vector<int64_t> tbase((size_t)nt + 1, 0);
    for (int t = 0; t < nt; ++t) tbase[t + 1] = tbase[t] + tg->len[t];
    const int64_t total = tbase[nt];
    telr_consensus *C = new telr_consensus();
    C->off.assign((size_t)nt, 0); C->len.assign((size_t)nt, 0);
    if (total == 0) { *out = C; return TELR_OK; }
    telr_aln *d_alns; uint32_t *d_cig = nullptr, *d_cell; int64_t *d_tbase, *d_e64, *d_woff, *d_hw; int32_t *d_emit; uint8_t *d_out;
    int rc;
    auto fail = [&](int code) { delete C; return code; };
    // the CIGAR array: the result's own device copy when it kept one (TELR_MF_KEEP_CIGARS; round 5: polish_consensus asks for it --
    // the upload was 250 MB of pageable memory per 1,000 configs[2] loci), else uploaded
    const bool twin = r->d_cig && !r->twin_off && r->twin_n == r->ncig;
    if (twin) { d_cig = r->d_cig; if (hipDeviceSynchronize() != hipSuccess) return fail(TELR_E_HIP); }      // (its last pieces were copied on other streams, as in telr_depth_medians)
    if ((rc = ctx_buf_t(ctx, "cons_alns", n + 1, &d_alns)) != TELR_OK || (!twin && (rc = ctx_buf_t(ctx, "cons_cig", r->ncig + 1, &d_cig)) != TELR_OK) ||
        (rc = ctx_buf_t(ctx, "cons_hw", (size_t)nt + 1, &d_hw)) != TELR_OK ||
        (rc = ctx_buf_t(ctx, "cons_cell", (size_t)total * CONS_CELL, &d_cell)) != TELR_OK || (rc = ctx_buf_t(ctx, "cons_tbase", (size_t)nt + 1, &d_tbase)) != TELR_OK ||
        (rc = ctx_buf_t(ctx, "cons_emit", (size_t)total + 1, &d_emit)) != TELR_OK || (rc = ctx_buf_t(ctx, "cons_e64", (size_t)total + 1, &d_e64)) != TELR_OK ||
        (rc = ctx_buf_t(ctx, "cons_woff", (size_t)total + 1, &d_woff)) != TELR_OK) return fail(rc);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ctx->err = std::string(#x) + ": " + hipGetErrorString(e_); return fail(e_ == hipErrorOutOfMemory ? TELR_E_NOMEM : TELR_E_HIP); } } while (0)
    if (n) CK(hipMemcpyAsync(d_alns, r->alns.data(), n * sizeof(telr_aln), hipMemcpyHostToDevice, st));
    if (r->ncig && !twin) CK(hipMemcpyAsync(d_cig, r->cig, r->ncig * 4, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(d_tbase, tbase.data(), ((size_t)nt + 1) * 8, hipMemcpyHostToDevice, st));
    CK(hipMemsetAsync(d_cell, 0, (size_t)total * CONS_CELL * 4, st));
    if (n) hipLaunchKernelGGL(k_pile_count, dim3((unsigned)n), dim3(64), 0, st, d_alns, (int32_t)n, d_cig, queries->d_seq2, queries->d_nmask, queries->d_boff, d_tbase, d_cell);
    const unsigned nb = (unsigned)((total + 256) / 256);
    hipLaunchKernelGGL(k_pile_call<0>, dim3(nb), dim3(256), 0, st, d_cell, total, d_tbase, nt, tg->d_seq2, tg->d_nmask, tg->d_boff, min_depth, d_emit, (const int64_t*)nullptr, (uint8_t*)nullptr);
    hipLaunchKernelGGL(k_widen_i32, dim3(nb), dim3(256), 0, st, d_emit, total + 1, d_e64);
    CK(hipGetLastError());
    if ((rc = dev_exclusive_scan<int64_t, int64_t>(ctx, d_e64, d_woff, (size_t)total + 1)) != TELR_OK) return fail(rc);
    // offsets of the targets in the output = the scan at their first position
    std::vector<int64_t> h_w((size_t)nt + 1);
    hipLaunchKernelGGL(k_pick_i64, dim3((unsigned)((nt + 256) / 256)), dim3(256), 0, st, d_woff, d_tbase, nt + 1, d_hw);      // (one copy, not one per target)
    CK(hipGetLastError());
    CK(hipMemcpyAsync(h_w.data(), d_hw, ((size_t)nt + 1) * 8, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    const int64_t wtot = h_w[nt];
    if ((rc = ctx_buf_t(ctx, "cons_out", (size_t)wtot + 1, &d_out)) != TELR_OK) return fail(rc);
    hipLaunchKernelGGL(k_pile_call<1>, dim3(nb), dim3(256), 0, st, d_cell, total, d_tbase, nt, tg->d_seq2, tg->d_nmask, tg->d_boff, min_depth, d_emit, d_woff, d_out);
    CK(hipGetLastError());
    C->seq.resize((size_t)wtot);
    if (wtot) CK(hipMemcpyAsync(&C->seq[0], d_out, (size_t)wtot, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
#undef CK
    for (int t = 0; t < nt; ++t) { C->off[t] = h_w[t]; C->len[t] = (int32_t)(h_w[t + 1] - h_w[t]); }
    *out = C;
    return TELR_OK;
}
