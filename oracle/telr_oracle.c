/*
 * telr_oracle.c — CPU restatement of the TELR alignment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in telr_amd/ may import, link or call this
 * file; it is used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the checker for the HIP engine (telr_amd/csrc).
 *
 * PARITY STATUS: "parity unpinned" against the upstream binaries.  The
 * arithmetic of this path lives in third-party tools that are absent from
 * /root/reference and from this image: minimap2 2.22 (envs/telr.yml:45; call
 * sites src/telr/TELR_alignment.py:69-82, TELR_assembly.py:199-212,
 * TELR_te.py:68-78,119-132,504-506, TELR_liftover.py:253-266) and ngmlr 0.2.7
 * (envs/telr.yml:48; TELR_alignment.py:31-51).  The reference ships no tests
 * or golden vectors for them.  This file restates the PUBLISHED algorithm
 * (Li 2018, Bioinformatics 34:3094, sections 2.1-2.3: (w,k)-minimizers with an
 * invertible hash, seed collection with an occurrence filter, anchor chaining
 * f(i)=max{max_j f(j)+alpha(j,i)-beta(j,i), w_i}, two-piece affine gap DP
 * between anchors with z-drop end extension, mapQ=40(1-f2/f1)min{1,m/10}log f1;
 * Li 2021, Bioinformatics 37:4572 for the long-read presets) with every
 * tie-break and heuristic fixed explicitly (see DESIGN.md "Engine spec") so a
 * GPU implementation can be bit-exact against it.  It is pinned by (i) the
 * reference's own bundled fixture (test/ FASTA files -> one jockey insertion near
 * ref offset 33017, minus strand; tests/test_fixture_known_answer.py) and
 * (ii) simulated-truth recovery.  The Python glue (liftover, AF) is pinned
 * separately against golden vectors captured from the reference's own code
 * (tests/golden/, tools/capture_goldens.py).
 *
 * Build: make -C oracle   ->  oracle/libtelroracle.so
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include "../include/telr_hip.h"

#define NEG        (-(1 << 28))
#define TPAD       16384          /* padding between targets in global coordinates */
#define KEY_REV    (1ULL << 63)

/* Oracle-only experiment bits (with TELR_MF_FAITHFUL they quantify what the engine's spec leaves out of minimap2 2.22's
 * published behaviour; tests/test_faithful_gate.py prints the drift table, DESIGN.md section 2 holds it):
 *   0x100  look-back 5000 (max_chain_iter)          0x200  gap fills over the whole -r band     0x400  uncapped end extensions
 *   0x1000 minimap2's order-dependent predecessor scan: look-back 5000, the scan starts where the reference distance exceeds
 *          max_gap, and stops after max_chain_skip = 25 predecessors that already sit on a chain through anchor i
 *          (lchain.c: mg_lchain_dp; Li 2018 section 2.1.1 "heuristics")
 *   0x2000 high-occurrence seed rescue: in a stretch of skipped (too frequent) minimizers longer than 500 query bases the
 *          least frequent ones (one per 500 bases, occurrence < 4096) are seeded after all (seed.c: mm_seed_select)
 *   0x4000 long join: when the first round leaves more than one chain, the anchors are chained again with the long bandwidth
 *          bw_long = 20,000 diagonals (map.c: the re-chaining of 2.19+; Li 2021)
 *   0x8000 RMQ chaining of the asm presets: -r100k with an unbounded look-back (what a range-minimum query sees)
 *   0x10000 z-drop inside gap fills: minimap2 breaks a record where the score of a fill's path drops by more than -z below
 *          its running maximum; here such fills are COUNTED per record (telr_aln.n_ambi) -- the records minimap2 would split
 *   0x20000 minimap2's own MAPQ (identity- and n_sub-aware; map.c: mm_set_mapq) instead of the paper's formula */
#define MFX_LOOKBACK 0x100
#define MFX_SKIP     0x1000
#define MFX_RESCUE   0x2000
#define MFX_LONGJOIN 0x4000
#define MFX_RMQ      0x8000
#define MFX_ZSPLIT   0x10000
#define MFX_MAPQ     0x20000
#define MFX_CONVEX   0x80000      /* ngmlr-* presets: NGMLR's convex gap cost exactly (length-tracking cells, scores in 1/20 units) instead of the two-piece envelope */
#define DP_DMAX    4096           /* widest band (diagonals) the DP accepts; wider -> diagonal fallback */
#define DEPTH_CAP  8000           /* samtools depth default -d */

/* ------------------------------------------------------------------------- */
/* growable arrays                                                            */
#define VEC(T) struct { T *a; int64_t n, m; }
#define vpush(T, v, x) do { if ((v).n == (v).m) { (v).m = (v).m ? (v).m * 2 : 64; \
        (v).a = (T*)realloc((v).a, sizeof(T) * (v).m); } (v).a[(v).n++] = (x); } while (0)

static const uint8_t NT4[256] = {
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4,4,4,4,4,3,3,4,4,4,4,4,4,4,4,4,4,
    4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4,4,4,4,4,3,3,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
};

uint8_t tor_nt4(uint8_t c) { return NT4[c]; }

/* invertible integer hash (Thomas Wang), Li 2018 section 2.1 */
static inline uint64_t hash64(uint64_t key, uint64_t mask)
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

/* ------------------------------------------------------------------------- */
/* 1. minimizer sketch                                                        */
/* A minimizer is (x, y): x = hash<<8 | span, y = (base + pos)<<1 | strand,   */
/* pos = last base of the k-mer.  Set definition (Li 2018 Alg. 1 without the  */
/* streaming quirks): slot u is selected iff it is valid and x_u <= x_v for   */
/* every v of SOME window of min(w, nslots) consecutive slots containing u.   */
typedef struct { uint64_t x; uint32_t y; } mz_t;
typedef VEC(mz_t) mzv_t;

static void sketch(const uint8_t *s /* nt4 codes */, int len, int k, int w, int hpc,
                   uint32_t base, mzv_t *out)
{
    if (len < k) return;
    int nu = 0;
    uint8_t *hb; int32_t *hs, *he;
    hb = (uint8_t*)malloc(len); hs = (int32_t*)malloc(4 * (size_t)len); he = (int32_t*)malloc(4 * (size_t)len);
    if (hpc) {
        for (int i = 0; i < len; ) {
            int j = i + 1;
            while (j < len && s[j] == s[i]) ++j;
            hb[nu] = s[i]; hs[nu] = i; he[nu] = j - 1; ++nu; i = j;
        }
    } else {
        for (int i = 0; i < len; ++i) { hb[i] = s[i]; hs[i] = he[i] = i; }
        nu = len;
    }
    int ns = nu - k + 1;
    if (ns >= 1) {
        uint64_t mask = (1ULL << 2 * k) - 1, shift1 = 2 * (k - 1);
        uint64_t *x = (uint64_t*)malloc(8 * (size_t)ns);
        uint32_t *y = (uint32_t*)malloc(4 * (size_t)ns);
        uint64_t fw = 0, rv = 0; int l = 0;
        for (int i = 0; i < nu; ++i) {
            int c = hb[i];
            if (c < 4) { fw = (fw << 2 | c) & mask; rv = (rv >> 2) | (3ULL ^ c) << shift1; ++l; }
            else l = 0, fw = rv = 0;
            if (i >= k - 1) {
                int u = i - k + 1;
                int span = he[i] - hs[u] + 1;
                if (l >= k && fw != rv && span < 256) {
                    int z = fw < rv ? 0 : 1;
                    x[u] = hash64(z ? rv : fw, mask) << 8 | (uint64_t)span;
                    y[u] = (base + (uint32_t)he[i]) << 1 | (uint32_t)z;
                } else x[u] = UINT64_MAX, y[u] = 0;
            }
        }
        int need = w < ns ? w : ns;
        for (int u = 0; u < ns; ++u) {
            if (x[u] == UINT64_MAX) continue;
            int L = 0, R = 0;
            while (L < w - 1 && u - L - 1 >= 0 && x[u - L - 1] >= x[u]) ++L;
            while (R < w - 1 && u + R + 1 < ns && x[u + R + 1] >= x[u]) ++R;
            if (L + R + 1 >= need) { mz_t m = { x[u], y[u] }; vpush(mz_t, *out, m); }
        }
        free(x); free(y);
    }
    free(hb); free(hs); free(he);
}

/* debug entry: sketch one ASCII sequence */
int64_t tor_sketch(const char *ascii, int32_t len, int k, int w, int hpc, uint32_t base,
                   uint64_t *x_out, uint32_t *y_out, int64_t cap)
{
    uint8_t *s = (uint8_t*)malloc(len > 0 ? len : 1);
    for (int i = 0; i < len; ++i) s[i] = NT4[(uint8_t)ascii[i]];
    mzv_t v = {0, 0, 0};
    sketch(s, len, k, w, hpc, base, &v);
    int64_t n = v.n;
    for (int64_t i = 0; i < n && i < cap; ++i) { x_out[i] = v.a[i].x; y_out[i] = v.a[i].y; }
    free(v.a); free(s);
    return n;
}

/* ------------------------------------------------------------------------- */
/* 2. index                                                                   */
typedef struct tor_index {
    int32_t k, w, hpc;
    int32_t n_seq;
    uint8_t **seq;        /* nt4 codes per target */
    int32_t *len;
    uint32_t *goff;       /* global coordinate of base 0 of each target */
    int64_t n_mz;
    uint64_t *hash;       /* [n_mz] sorted by (hash, y) */
    uint32_t *ys;         /* [n_mz] */
    int64_t n_ent;
    uint64_t *ent_hash;   /* [n_ent] distinct hashes ascending */
    uint32_t *ent_off;    /* [n_ent+1] */
    /* per target: the occurrence counts of its own distinct minimizers, ascending (for the per-target cut-off) */
    uint32_t **pt_cnt; int64_t *pt_n;
} tor_index;

typedef struct { uint64_t h; uint32_t y; } hy_t;
static int cmp_hy(const void *a, const void *b)
{
    const hy_t *p = (const hy_t*)a, *q = (const hy_t*)b;
    if (p->h != q->h) return p->h < q->h ? -1 : 1;
    return p->y < q->y ? -1 : p->y > q->y;
}

static inline int32_t tid_of_gpos(const tor_index *ix, uint32_t g)
{
    int32_t lo = 0, hi = ix->n_seq - 1;
    while (lo < hi) { int32_t mid = (lo + hi + 1) >> 1; if (ix->goff[mid] <= g) lo = mid; else hi = mid - 1; }
    return lo;
}
static int cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b; return x < y ? -1 : x > y; }

tor_index *tor_index_build(int32_t n, const char *ascii, const int64_t *off, const int32_t *len,
                           const telr_idx_opt *io)
{
    tor_index *ix = (tor_index*)calloc(1, sizeof(*ix));
    ix->k = io->k; ix->w = io->w; ix->hpc = io->is_hpc; ix->n_seq = n;
    ix->seq = (uint8_t**)calloc(n, sizeof(uint8_t*));
    ix->len = (int32_t*)calloc(n, 4); ix->goff = (uint32_t*)calloc(n + 1, 4);
    mzv_t v = {0, 0, 0};
    uint64_t g = 0;
    for (int i = 0; i < n; ++i) {
        ix->len[i] = len[i]; ix->goff[i] = (uint32_t)g;
        ix->seq[i] = (uint8_t*)malloc(len[i] > 0 ? len[i] : 1);
        for (int j = 0; j < len[i]; ++j) ix->seq[i][j] = NT4[(uint8_t)ascii[off[i] + j]];
        sketch(ix->seq[i], len[i], io->k, io->w, io->is_hpc, (uint32_t)g, &v);
        g = (g + (uint64_t)len[i] + TPAD + 63) & ~63ULL;
    }
    ix->goff[n] = (uint32_t)g;
    hy_t *t = (hy_t*)malloc(sizeof(hy_t) * (v.n ? v.n : 1));
    for (int64_t i = 0; i < v.n; ++i) { t[i].h = v.a[i].x >> 8; t[i].y = v.a[i].y; }
    /* NOTE the span byte is not part of the index order: grouping is by hash only */
    if (v.n) qsort(t, v.n, sizeof(hy_t), cmp_hy);
    ix->n_mz = v.n;
    ix->hash = (uint64_t*)malloc(8 * (size_t)(v.n ? v.n : 1)); ix->ys = (uint32_t*)malloc(4 * (size_t)(v.n ? v.n : 1));
    int64_t ne = 0;
    for (int64_t i = 0; i < v.n; ++i) { ix->hash[i] = t[i].h; ix->ys[i] = t[i].y; if (i == 0 || t[i].h != t[i-1].h) ++ne; }
    ix->n_ent = ne;
    ix->ent_hash = (uint64_t*)malloc(8 * (ne ? ne : 1)); ix->ent_off = (uint32_t*)malloc(4 * (ne + 1));
    ne = 0;
    for (int64_t i = 0; i < v.n; ++i) if (i == 0 || t[i].h != t[i-1].h) { ix->ent_hash[ne] = t[i].h; ix->ent_off[ne++] = (uint32_t)i; }
    ix->ent_off[ne] = (uint32_t)v.n;
    free(t); free(v.a);
    /* per-target counts: inside one hash group the positions ascend, i.e. they are grouped by target */
    ix->pt_cnt = (uint32_t**)calloc(n ? n : 1, sizeof(uint32_t*)); ix->pt_n = (int64_t*)calloc(n ? n : 1, 8);
    int64_t *cap = (int64_t*)calloc(n ? n : 1, 8);
    for (int64_t e = 0; e < ix->n_ent; ++e) {
        uint32_t o = ix->ent_off[e], o1 = ix->ent_off[e + 1];
        while (o < o1) {
            int32_t tid = tid_of_gpos(ix, ix->ys[o] >> 1);
            uint32_t z = o + 1;
            while (z < o1 && (ix->ys[z] >> 1) < ix->goff[tid + 1]) ++z;
            if (ix->pt_n[tid] == cap[tid]) { cap[tid] = cap[tid] ? cap[tid] * 2 : 64; ix->pt_cnt[tid] = (uint32_t*)realloc(ix->pt_cnt[tid], 4 * cap[tid]); }
            ix->pt_cnt[tid][ix->pt_n[tid]++] = z - o;
            o = z;
        }
    }
    for (int i = 0; i < n; ++i) if (ix->pt_n[i]) qsort(ix->pt_cnt[i], ix->pt_n[i], 4, cmp_u32);
    free(cap);
    return ix;
}

void tor_index_free(tor_index *ix)
{
    if (!ix) return;
    for (int i = 0; i < ix->n_seq; ++i) free(ix->seq[i]);
    for (int i = 0; i < ix->n_seq; ++i) free(ix->pt_cnt[i]);
    free(ix->pt_cnt); free(ix->pt_n);
    free(ix->seq); free(ix->len); free(ix->goff); free(ix->hash); free(ix->ys); free(ix->ent_hash); free(ix->ent_off); free(ix);
}

int64_t tor_index_n_mz(const tor_index *ix) { return ix->n_mz; }
int64_t tor_index_n_ent(const tor_index *ix) { return ix->n_ent; }
void tor_index_dump(const tor_index *ix, uint64_t *hash, uint32_t *ys)
{
    memcpy(hash, ix->hash, 8 * ix->n_mz); memcpy(ys, ix->ys, 4 * ix->n_mz);
}

/* occurrence cut-off: the (1-f) quantile of the distinct-minimizer counts, +1, clamped */
int32_t tor_mid_occ(const tor_index *ix, float frac, int32_t lo, int32_t hi)
{
    int64_t n = ix->n_ent;
    int32_t occ;
    if (n == 0) occ = lo;
    else {
        uint32_t *c = (uint32_t*)malloc(4 * n);
        for (int64_t i = 0; i < n; ++i) c[i] = ix->ent_off[i + 1] - ix->ent_off[i];
        qsort(c, n, 4, cmp_u32);
        int64_t idx = (int64_t)((1.0 - (double)frac) * (double)n);
        if (idx >= n) idx = n - 1;
        occ = (int32_t)c[idx] + 1;
        free(c);
    }
    if (occ < lo) occ = lo;
    if (hi > lo && occ > hi) occ = hi;
    return occ;
}

/* the same rule on ONE target's own minimizers: what an aligner run against that target alone would use (the
 * reference runs minimap2 once per contig at the per-locus sites, TELR_te.py:68-78,119-132,504-506) */
int32_t tor_mid_occ_target(const tor_index *ix, int32_t t, float frac, int32_t lo, int32_t hi)
{
    int64_t n = ix->pt_n[t];
    int32_t occ;
    if (n == 0) occ = lo;
    else {
        int64_t idx = (int64_t)((1.0 - (double)frac) * (double)n);
        if (idx >= n) idx = n - 1;
        occ = (int32_t)ix->pt_cnt[t][idx] + 1;
    }
    if (occ < lo) occ = lo;
    if (hi > lo && occ > hi) occ = hi;
    return occ;
}

static inline int64_t index_lookup(const tor_index *ix, uint64_t h)
{
    int64_t lo = 0, hi = ix->n_ent - 1;
    while (lo <= hi) {
        int64_t mid = (lo + hi) >> 1;
        if (ix->ent_hash[mid] == h) return mid;
        if (ix->ent_hash[mid] < h) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

/* ------------------------------------------------------------------------- */
/* 3. seeding: anchors as sortable 64-bit keys                                */
/*   bit 63     : query strand differs from target strand                     */
/*   bits 62-32 : global target coordinate of the k-mer's last base           */
/*   bits 31-8  : query coordinate (on the strand-adjusted query) of the last base */
/*   bits 7-0   : query span                                                  */
typedef VEC(uint64_t) u64v_t;
static int cmp_u64(const void *a, const void *b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }

#define A_REV(k)  ((int)((k) >> 63))
#define A_G(k)    ((int32_t)(((k) >> 32) & 0x7fffffff))
#define A_Q(k)    ((int32_t)(((k) >> 8) & 0xffffff))
#define A_SPAN(k) ((int32_t)((k) & 0xff))

/* Sub-read voting (spec 3.10; NGMLR's candidate search, Sedlazeck 2018 / NextGenMap: SURVEY App. A.6).  hits[i0..i1) are the
 * hits of the minimizers of ONE sub-read.  Every hit votes for its diagonal bin (reference minus strand-adjusted query
 * position, 2^vote_bin_shift bases wide, 1024 bins per strand, folded); a hit stays iff its bin and the two neighbours hold
 * at least max(vote_min, ceil(vote_frac_q8/256 x votes of the fullest bin)) hits. */
#define VOTE_SLOTS 2048
static inline uint32_t vote_slot(uint64_t key, int shift)
{
    uint32_t d = (uint32_t)A_G(key) - (uint32_t)A_Q(key) + (1u << 24);
    return (((d >> shift) & (VOTE_SLOTS / 2 - 1)) << 1) | (uint32_t)A_REV(key);
}
static void vote_subread(const uint64_t *hits, int64_t n, const telr_map_opt *mo, u64v_t *out)
{
    uint32_t tab[VOTE_SLOTS];
    memset(tab, 0, sizeof(tab));
    uint32_t v1 = 0;
    for (int64_t i = 0; i < n; ++i) { uint32_t c = ++tab[vote_slot(hits[i], mo->vote_bin_shift)]; if (c > v1) v1 = c; }
    uint32_t thr = (v1 * (uint32_t)mo->vote_frac_q8 + 255u) >> 8;
    if (thr < (uint32_t)mo->vote_min) thr = (uint32_t)mo->vote_min;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t s = vote_slot(hits[i], mo->vote_bin_shift);
        uint32_t w = tab[(s - 2) & (VOTE_SLOTS - 1)] + tab[s] + tab[(s + 2) & (VOTE_SLOTS - 1)];
        if (w >= thr) vpush(uint64_t, *out, hits[i]);
    }
}

static void collect_anchors(const tor_index *ix, const uint8_t *q, int qlen, int32_t tfilter, int all_vs_all, int32_t mid_occ, const telr_map_opt *mo,
                            u64v_t *out, int64_t *n_mz, int64_t *n_probe)
{
    mzv_t mv = {0, 0, 0};
    sketch(q, qlen, ix->k, ix->w, ix->hpc, 0, &mv);
    *n_mz += mv.n;
    const int per_t = tfilter < 0 && (mo->flags & TELR_MF_PER_TARGET);
    /* spec 3.10: sub-read voting is the candidate search of ALL-VS-ALL calls -- a call that carries a per-query target array is
     * not one, whatever the entry of this query says (-1 = unrestricted) */
    const int vote = mo->vote_len > 0 && all_vs_all && !per_t;
    u64v_t sub = {0, 0, 0};          /* hits of the sub-read in progress */
    int32_t sub_id = -1;
    uint32_t g0 = 0, g1 = 0xffffffffu;
    if (tfilter >= 0) {
        g0 = ix->goff[tfilter]; g1 = g0 + (uint32_t)ix->len[tfilter];
        mid_occ = tor_mid_occ_target(ix, tfilter, mo->mid_occ_frac, mo->min_mid_occ, mo->max_mid_occ);
    }
    /* 0x2000: which of the too-frequent minimizers are seeded after all (mm_seed_select; all-vs-all calls only) */
    uint8_t *rescued = NULL;
    if ((mo->flags & MFX_RESCUE) && tfilter < 0 && !per_t && mv.n > 0) {
        const int dist = 500, max_max_occ = 4095;
        rescued = (uint8_t*)calloc(mv.n, 1);
        int64_t *occ = (int64_t*)malloc(8 * mv.n);
        for (int64_t i = 0; i < mv.n; ++i) { int64_t e = index_lookup(ix, mv.a[i].x >> 8); occ[i] = e < 0 ? 0 : (int64_t)(ix->ent_off[e + 1] - ix->ent_off[e]); }
        int64_t last0 = -1;
        for (int64_t i = 0; i <= mv.n; ++i) {
            if (i == mv.n || occ[i] <= mid_occ) {
                if (i - last0 > 1) {
                    int32_t ps = last0 < 0 ? 0 : (int32_t)(mv.a[last0].y >> 1), pe = i == mv.n ? qlen : (int32_t)(mv.a[i].y >> 1);
                    int k = (int)((double)(pe - ps) / dist + .499);
                    /* the k least frequent of the stretch (ties: the earlier one), below max_max_occ */
                    for (int z = 0; z < k; ++z) {
                        int64_t bi = -1;
                        for (int64_t j = last0 + 1; j < i; ++j) if (!rescued[j] && occ[j] < max_max_occ && (bi < 0 || occ[j] < occ[bi])) bi = j;
                        if (bi < 0) break;
                        rescued[bi] = 1;
                    }
                }
                last0 = i;
            }
        }
        free(occ);
    }
    for (int64_t i = 0; i < mv.n; ++i) {
        int64_t e = index_lookup(ix, mv.a[i].x >> 8);
        ++*n_probe;
        if (e < 0) continue;
        uint32_t o0 = ix->ent_off[e], o1 = ix->ent_off[e + 1];
        int32_t span = (int32_t)(mv.a[i].x & 0xff), qpos = (int32_t)(mv.a[i].y >> 1), qz = (int32_t)(mv.a[i].y & 1);
        if (per_t) {
            /* one run of occurrences per target; each run is tested against its target's own cut-off */
            for (uint32_t o = o0; o < o1; ) {
                int32_t tid = tid_of_gpos(ix, ix->ys[o] >> 1);
                uint32_t z = o + 1;
                while (z < o1 && (ix->ys[z] >> 1) < ix->goff[tid + 1]) ++z;
                if ((int32_t)(z - o) <= tor_mid_occ_target(ix, tid, mo->mid_occ_frac, mo->min_mid_occ, mo->max_mid_occ))
                    for (uint32_t y = o; y < z; ++y) {
                        uint32_t g = ix->ys[y] >> 1; int tz = ix->ys[y] & 1;
                        uint64_t key = tz == qz ? (uint64_t)g << 32 | (uint64_t)qpos << 8 | (uint64_t)span
                                                : KEY_REV | (uint64_t)g << 32 | (uint64_t)(qlen - (qpos + 1 - span) - 1) << 8 | (uint64_t)span;
                        vpush(uint64_t, *out, key);
                    }
                o = z;
            }
            continue;
        }
        int32_t cnt = 0;
        if (tfilter >= 0) { for (uint32_t o = o0; o < o1; ++o) { uint32_t g = ix->ys[o] >> 1; if (g >= g0 && g < g1) ++cnt; } }
        else cnt = (int32_t)(o1 - o0);
        if (cnt == 0 || (cnt > mid_occ && !(rescued && rescued[i]))) continue;
        if (vote) {
            /* minimizers come in query order: a new sub-read closes the previous one */
            int32_t sid = qpos / mo->vote_len;
            if (sid != sub_id) { vote_subread(sub.a, sub.n, mo, out); sub.n = 0; sub_id = sid; }
            for (uint32_t o = o0; o < o1; ++o) {
                uint32_t g = ix->ys[o] >> 1; int tz = ix->ys[o] & 1;
                uint64_t key = tz == qz ? (uint64_t)g << 32 | (uint64_t)qpos << 8 | (uint64_t)span
                                        : KEY_REV | (uint64_t)g << 32 | (uint64_t)(qlen - (qpos + 1 - span) - 1) << 8 | (uint64_t)span;
                vpush(uint64_t, sub, key);
            }
            continue;
        }
        for (uint32_t o = o0; o < o1; ++o) {
            uint32_t g = ix->ys[o] >> 1;
            if (tfilter >= 0 && (g < g0 || g >= g1)) continue;
            int tz = ix->ys[o] & 1;
            uint64_t key;
            if (tz == qz) key = (uint64_t)g << 32 | (uint64_t)qpos << 8 | (uint64_t)span;
            else key = KEY_REV | (uint64_t)g << 32 | (uint64_t)(qlen - (qpos + 1 - span) - 1) << 8 | (uint64_t)span;
            vpush(uint64_t, *out, key);
        }
    }
    if (vote) { vote_subread(sub.a, sub.n, mo, out); free(sub.a); }
    free(rescued);
    free(mv.a);
    if (out->n) qsort(out->a, out->n, 8, cmp_u64);
}

/* ------------------------------------------------------------------------- */
/* 4. chaining                                                                */
/* integer log2 in Q8 (piecewise-linear mantissa); v >= 1                     */
static inline int32_t ilog2_q8(uint32_t v)
{
    int e = 31 - __builtin_clz(v);
    uint32_t frac = e <= 8 ? (v << (8 - e)) - 256u : (v >> (e - 8)) - 256u;
    return e * 256 + (int32_t)frac;
}

/* score of appending anchor i after anchor j; INT32_MIN if not allowed */
static inline int32_t chain_sc(uint64_t ai, uint64_t aj, const telr_map_opt *mo)
{
    if ((ai >> 63) != (aj >> 63)) return INT32_MIN;
    int32_t dr = A_G(ai) - A_G(aj), dq = A_Q(ai) - A_Q(aj);
    if (dq <= 0 || dq > mo->max_gap) return INT32_MIN;
    if (dr <= 0 || dr > mo->max_gap) return INT32_MIN;
    int32_t dd = dr > dq ? dr - dq : dq - dr;
    if (dd > (mo->bw_long > mo->bw ? mo->bw_long : mo->bw)) return INT32_MIN;      /* long join: chained within max(bw, bw_long) diagonals */
    int32_t dg = dr < dq ? dr : dq;
    int32_t span = A_SPAN(ai);
    int32_t sc = span < dg ? span : dg;
    if (dd || dg > span) {
        int32_t pen = mo->chain_gap_q8 * dd + mo->chain_skip_q8 * dg + (dd >= 1 ? ilog2_q8((uint32_t)dd + 1) >> 1 : 0);
        sc -= pen >> 8;
    }
    return sc;
}

static void chain_dp(const uint64_t *a, int64_t n, const telr_map_opt *mo, int32_t *f, int32_t *p)
{
    int H = (mo->flags & (TELR_MF_FAITHFUL | MFX_LOOKBACK | MFX_SKIP)) ? 5000 : mo->chain_lookback;     /* faithful mode: minimap2's max_chain_iter */
    if (mo->flags & MFX_RMQ) H = 1 << 30;
    if (mo->flags & MFX_SKIP) {
        const int max_skip = 25;
        int32_t *t = (int32_t*)malloc(4 * (n ? n : 1));
        for (int64_t i = 0; i < n; ++i) t[i] = -1;
        int64_t st = 0;
        for (int64_t i = 0; i < n; ++i) {
            while (st < i && ((a[st] >> 63) != (a[i] >> 63) || A_G(a[i]) - A_G(a[st]) > mo->max_gap)) ++st;
            if (i - st > H) st = i - H;
            int32_t best = A_SPAN(a[i]), bp = -1; int n_skip = 0;
            for (int64_t j = i - 1; j >= st; --j) {
                int32_t sc = chain_sc(a[i], a[j], mo);
                if (sc == INT32_MIN) continue;
                int32_t v = f[j] + sc;
                if (v > best) { best = v; bp = (int32_t)j; if (n_skip > 0) --n_skip; }
                else if (t[j] == (int32_t)i) { if (++n_skip > max_skip) break; }
                if (p[j] >= 0) t[p[j]] = (int32_t)i;
            }
            f[i] = best; p[i] = bp;
        }
        free(t);
        return;
    }
    for (int64_t i = 0; i < n; ++i) {
        int32_t best = A_SPAN(a[i]), bp = -1;
        int64_t st = i - H; if (st < 0) st = 0;
        for (int64_t j = i - 1; j >= st; --j) {
            int32_t sc = chain_sc(a[i], a[j], mo);
            if (sc == INT32_MIN) continue;
            int32_t v = f[j] + sc;
            if (v > best) best = v, bp = (int32_t)j;
        }
        f[i] = best; p[i] = bp;
    }
}

typedef struct {
    int32_t score, cnt;
    int64_t a_off;          /* into the per-query chain-anchor array */
    int32_t rev, tid;
    int32_t rs, re, qs, qe; /* target-local / strand-adjusted query coordinates */
    int32_t disc;           /* discovery order */
} chain_t;
typedef VEC(chain_t) chainv_t;

typedef struct { int32_t f, i; } peak_t;
static int cmp_peak(const void *a, const void *b)
{
    const peak_t *x = (const peak_t*)a, *y = (const peak_t*)b;
    if (x->f != y->f) return x->f > y->f ? -1 : 1;
    return x->i < y->i ? -1 : x->i > y->i;
}

/* back-tracking: start from peaks (no successor with a larger f) in (f desc, index asc) order */
static void chain_backtrack(const tor_index *ix, const uint64_t *a, int64_t n, const int32_t *f, const int32_t *p,
                            const telr_map_opt *mo, chainv_t *chains, u64v_t *canch)
{
    uint8_t *nonpeak = (uint8_t*)calloc(n ? n : 1, 1), *vis = (uint8_t*)calloc(n ? n : 1, 1);
    for (int64_t i = 0; i < n; ++i) if (p[i] >= 0 && f[i] > f[p[i]]) nonpeak[p[i]] = 1;
    VEC(peak_t) pk = {0, 0, 0};
    for (int64_t i = 0; i < n; ++i) if (!nonpeak[i] && f[i] >= mo->min_chain_score) { peak_t t = { f[i], (int32_t)i }; vpush(peak_t, pk, t); }
    if (pk.n) qsort(pk.a, pk.n, sizeof(peak_t), cmp_peak);      /* (an empty vector was never allocated: qsort's pointer must not be null) */
    for (int64_t t = 0; t < pk.n; ++t) {
        int32_t i = pk.a[t].i;
        if (vis[i]) continue;
        int32_t cnt = 0, j = i, stop_f = 0;
        while (j >= 0 && !vis[j]) { vis[j] = 1; ++cnt; j = p[j]; }
        if (j >= 0) stop_f = f[j];
        int32_t sc = f[i] - stop_f;
        if (sc < mo->min_chain_score || cnt < mo->min_cnt) continue;
        chain_t c; memset(&c, 0, sizeof(c));
        c.score = sc; c.cnt = cnt; c.a_off = canch->n; c.disc = (int32_t)chains->n;
        for (int32_t z = 0; z < cnt; ++z) vpush(uint64_t, *canch, 0);
        j = i;
        for (int32_t z = cnt - 1; z >= 0; --z) { canch->a[c.a_off + z] = a[j]; j = p[j]; }
        uint64_t a0 = canch->a[c.a_off], a1 = canch->a[c.a_off + cnt - 1];
        c.rev = A_REV(a0); c.tid = tid_of_gpos(ix, (uint32_t)A_G(a0));
        int32_t go = (int32_t)ix->goff[c.tid];
        c.rs = A_G(a0) - go - A_SPAN(a0) + 1; c.re = A_G(a1) - go + 1;
        /* the span is the QUERY minimizer's; with homopolymer compression the target's copy of the k-mer may be shorter and the start
         * fall before the target (minimap2 clamps the same way: mm_set_reg / mm_reg_set_coor, `x + 1 > q_span ? x + 1 - q_span : 0`) */
        if (c.rs < 0) c.rs = 0;
        c.qs = A_Q(a0) - A_SPAN(a0) + 1;      c.qe = A_Q(a1) + 1;
        vpush(chain_t, *chains, c);
    }
    free(nonpeak); free(vis); free(pk.a);
}

/* ------------------------------------------------------------------------- */
/* 5. chain selection (primary / secondary / supplementary)                   */
typedef struct {
    int32_t ci;            /* chain index */
    int32_t key;           /* sort key: chain score (pass 1) or dp score (pass 2) */
    int32_t ord;           /* tie-break: previous order */
    int32_t fs, fe;        /* query interval on the forward strand */
    int32_t tid;
    int32_t parent, subsc, n_sub, keep;
} sel_t;

static int cmp_sel(const void *a, const void *b)
{
    const sel_t *x = (const sel_t*)a, *y = (const sel_t*)b;
    if (x->key != y->key) return x->key > y->key ? -1 : 1;
    return x->ord < y->ord ? -1 : x->ord > y->ord;
}

/* s[] sorted; sets parent/subsc/n_sub/keep */
static void select_chains(sel_t *s, int n, const telr_map_opt *mo, const int32_t *sub_score /* by ci */)
{
    int per_t = (mo->flags & TELR_MF_PER_TARGET) != 0;
    for (int i = 0; i < n; ++i) {
        s[i].parent = i; s[i].subsc = 0; s[i].n_sub = 0;
        for (int j = 0; j < i; ++j) {
            if (s[j].parent != j) continue;
            if (per_t && s[j].tid != s[i].tid) continue;
            int32_t lo = s[i].fs > s[j].fs ? s[i].fs : s[j].fs, hi = s[i].fe < s[j].fe ? s[i].fe : s[j].fe;
            int32_t ol = hi > lo ? hi - lo : 0;
            int32_t li = s[i].fe - s[i].fs, lj = s[j].fe - s[j].fs, mn = li < lj ? li : lj;
            if ((float)ol > mo->mask_level * (float)mn) {
                s[i].parent = j;
                if (sub_score[s[i].ci] > s[j].subsc) s[j].subsc = sub_score[s[i].ci];
                ++s[j].n_sub;
                break;
            }
        }
    }
    /* keep secondaries scoring >= pri_ratio * parent, at most best_n (per query, or per target) */
    for (int i = 0; i < n; ++i) {
        if (s[i].parent == i) { s[i].keep = 1; continue; }
        s[i].keep = 0;
        if (!mo->secondary) continue;
        if ((float)s[i].key < (float)s[s[i].parent].key * mo->pri_ratio) continue;
        int n2 = 0;
        for (int j = 0; j < i; ++j) if (s[j].keep && s[j].parent != j && (!per_t || s[j].tid == s[i].tid)) ++n2;
        if (n2 < mo->best_n) s[i].keep = 1;
    }
}

/* ------------------------------------------------------------------------- */
/* 6. banded two-piece-affine DP over anti-diagonals                           */
/* Cell (i,j): i query bases, j target bases consumed; a=i+j, d=j-i.           */
/* Band dlo<=d<=dhi.  tb byte: bits0-2 source of H (0 diag,1 E1,2 F1,3 E2,4 F2) */
/* bit3 E1 extended, bit4 F1 extended, bit5 E2 extended, bit6 F2 extended.      */
typedef struct {
    const uint8_t *q, *t;   /* sequence accessors: base x is q[qi0 + qstep*x]            */
    int64_t qi0, ti0; int qstep, tstep; int qcomp; /* qcomp: complement the query base     */
    int m, n;
} dp_seq_t;

static inline int qbase(const dp_seq_t *s, int i) { int c = s->q[s->qi0 + (int64_t)s->qstep * i]; return (s->qcomp && c < 4) ? 3 - c : c; }
static inline int tbase(const dp_seq_t *s, int j) { return s->t[s->ti0 + (int64_t)s->tstep * j]; }

typedef VEC(uint32_t) u32v_t;
static inline void cig_push(u32v_t *c, int op, int len)
{
    if (len <= 0) return;
    if (c->n && (int)(c->a[c->n - 1] & 0xf) == op) c->a[c->n - 1] += (uint32_t)len << 4;
    else vpush(uint32_t, *c, (uint32_t)len << 4 | (uint32_t)op);
}

typedef struct { int score, bi, bj, touched; int64_t cells; } dp_res_t;

/* ext=0: global alignment of (m,n), returns H(m,n), traceback from (m,n).
 * ext=1: extension from (0,0): best cell with z-drop, traceback from it.
 * The CIGAR is appended to rev_cig in REVERSE order of ops (end -> start). */
/* (experiment 0x80000, below: the ngmlr-* presets' convex gap cost in exact form) */
typedef struct convex_s { int S, a, b, amb, open, emax, emin, dec, flat; } convex_t;
static int convex_of(const telr_map_opt *mo, convex_t *c)
{
    if (mo->cx_scale > 0) {                  /* the spec (telr_map_opt.cx_*): both ngmlr-* presets since round 4 */
        c->S = mo->cx_scale; c->a = mo->a * c->S; c->b = mo->b * c->S; c->amb = mo->sc_ambi * c->S;
        c->open = mo->cx_open; c->emax = mo->cx_ext_max; c->emin = mo->cx_ext_min; c->dec = mo->cx_decay;
        c->flat = c->dec > 0 && c->emax > c->emin ? (c->emax - c->emin + c->dec - 1) / c->dec : 0;
        return 1;
    }
    if (!(mo->flags & MFX_CONVEX)) return 0;
    if (mo->a == 2 && mo->b == 5 && mo->q == 6 && mo->e == 4 && mo->q2 == 60 && mo->e2 == 1) {          /* ngmlr-pacbio */
        c->S = 20; c->a = 40; c->b = 100; c->amb = 20 * mo->sc_ambi; c->open = 100; c->emax = 100; c->emin = 20; c->dec = 3; c->flat = 27; return 1;
    }
    if (mo->a == 2 && mo->b == 2 && mo->q == 2 && mo->e == 2 && mo->q2 == 4 && mo->e2 == 1) {           /* ngmlr-ont (the preset's units are already doubled) */
        c->S = 10; c->a = 20; c->b = 20; c->amb = 10 * mo->sc_ambi; c->open = 20; c->emax = 20; c->emin = 10; c->dec = 3; c->flat = 4; return 1;
    }
    return 0;
}
static inline int cx_ext(const convex_t *c, int len) { int v = c->emax - c->dec * len; return v > c->emin ? v : c->emin; }
static inline int64_t cx_cost(const convex_t *c, int L)       /* a whole gap of L bases */
{
    int64_t t = c->open;
    for (int i = 0; i < L && i < c->flat; ++i) t += cx_ext(c, i);
    if (L > c->flat) t += (int64_t)(L - c->flat) * c->emin;
    return t;
}
static dp_res_t band_dp_convex(const dp_seq_t *s, int dlo, int dhi, int ext, const telr_map_opt *mo, const struct convex_s *C, u32v_t *rev_cig);
static dp_res_t band_dp(const dp_seq_t *s, int dlo, int dhi, int ext, const telr_map_opt *mo, u32v_t *rev_cig)
{
    if (mo->cx_scale > 0 || (mo->flags & MFX_CONVEX)) { struct convex_s C; if (convex_of(mo, &C)) return band_dp_convex(s, dlo, dhi, ext, mo, &C, rev_cig); }
    const int m = s->m, n = s->n, D = dhi - dlo + 1, stride = (D + 2) / 2;
    const int q1 = mo->q, e1 = mo->e, q2 = mo->q2, e2 = mo->e2;
    dp_res_t res = { 0, 0, 0, 0, 0 };
    int32_t *H = (int32_t*)malloc(4 * (size_t)(D + 2) * 5), *E1 = H + (D + 2), *F1 = E1 + (D + 2), *E2 = F1 + (D + 2), *F2 = E2 + (D + 2);
    for (int x = 0; x < (D + 2) * 5; ++x) H[x] = NEG;
    uint8_t *tb = (uint8_t*)calloc((size_t)(m + n + 1) * stride, 1);
#define IX(d) ((d) - dlo + 1)
    int best = 0, bi = 0, bj = 0, prev_cur = NEG, last_a = m + n;
    if (0 >= dlo && 0 <= dhi) H[IX(0)] = 0;
    for (int a = 1; a <= m + n; ++a) {
        int d0 = -a > dlo ? -a : dlo; if (a - 2 * m > d0) d0 = a - 2 * m;
        int d1 = a < dhi ? a : dhi;   if (2 * n - a < d1) d1 = 2 * n - a;
        if (((d0 - a) & 1) != 0) ++d0;
        int cur = NEG, cur_d = 0;
        for (int d = d0; d <= d1; d += 2) {
            int i = (a - d) >> 1, j = (a + d) >> 1, x = IX(d);
            int32_t h, ve1, vf1, ve2, vf2; uint8_t t = 0;
            if (i == 0) {          /* first row: a deletion of length j */
                ve1 = -(q1 + j * e1); ve2 = -(q2 + j * e2); vf1 = vf2 = NEG;
                h = ve1 > ve2 ? ve1 : ve2;
            } else if (j == 0) {   /* first column: an insertion of length i */
                vf1 = -(q1 + i * e1); vf2 = -(q2 + i * e2); ve1 = ve2 = NEG;
                h = vf1 > vf2 ? vf1 : vf2;
            } else {
                int32_t hl = H[x - 1], hu = H[x + 1], hd = H[x];
                int32_t o, g;
                o = hl - q1 - e1; g = E1[x - 1] - e1; if (g > o) ve1 = g, t |= 8;  else ve1 = o;
                o = hu - q1 - e1; g = F1[x + 1] - e1; if (g > o) vf1 = g, t |= 16; else vf1 = o;
                o = hl - q2 - e2; g = E2[x - 1] - e2; if (g > o) ve2 = g, t |= 32; else ve2 = o;
                o = hu - q2 - e2; g = F2[x + 1] - e2; if (g > o) vf2 = g, t |= 64; else vf2 = o;
                int qb = qbase(s, i - 1), tbv = tbase(s, j - 1);
                int sc = (qb > 3 || tbv > 3) ? -mo->sc_ambi : (qb == tbv ? mo->a : -mo->b);
                h = hd + sc; int src = 0;
                if (ve1 > h) h = ve1, src = 1;
                if (vf1 > h) h = vf1, src = 2;
                if (ve2 > h) h = ve2, src = 3;
                if (vf2 > h) h = vf2, src = 4;
                t |= (uint8_t)src;
                tb[(size_t)a * stride + ((d - dlo) >> 1)] = t;
                ++res.cells;
            }
            /* clamp so that unreachable cells stay near NEG without drifting */
            if (h < NEG) h = NEG;
            if (ve1 < NEG) ve1 = NEG;
            if (vf1 < NEG) vf1 = NEG;
            if (ve2 < NEG) ve2 = NEG;
            if (vf2 < NEG) vf2 = NEG;
            H[x] = h; E1[x] = ve1; F1[x] = vf1; E2[x] = ve2; F2[x] = vf2;
            if (h > cur) cur = h, cur_d = d;
        }
        if (ext) {
            if (cur > best) best = cur, bi = (a - cur_d) >> 1, bj = (a + cur_d) >> 1;
            int c2 = cur > prev_cur ? cur : prev_cur;
            if (best - c2 > mo->zdrop) { last_a = a; break; }
            prev_cur = cur;
        }
    }
    (void)last_a;
    int i, j;
    if (ext) { res.score = best; i = bi; j = bj; }
    else { res.score = H[IX(n - m)]; i = m; j = n; }
    res.bi = i; res.bj = j;
    /* traceback */
    int state = 0;
    while (i > 0 && j > 0) {
        uint8_t t = tb[(size_t)(i + j) * stride + ((j - i - dlo) >> 1)];
        if (j - i - dlo <= mo->fill_margin || dhi - (j - i) <= mo->fill_margin) res.touched = 1;     /* the path used (nearly) all the slack of the band */
        if (state == 0) state = t & 7;
        if (state == 0) { cig_push(rev_cig, 0, 1); --i; --j; }
        else if (state == 1) { cig_push(rev_cig, 2, 1); if (!(t & 8))  state = 0; --j; }
        else if (state == 2) { cig_push(rev_cig, 1, 1); if (!(t & 16)) state = 0; --i; }
        else if (state == 3) { cig_push(rev_cig, 2, 1); if (!(t & 32)) state = 0; --j; }
        else                 { cig_push(rev_cig, 1, 1); if (!(t & 64)) state = 0; --i; }
    }
    if (i > 0) cig_push(rev_cig, 1, i);
    if (j > 0) cig_push(rev_cig, 2, j);
#undef IX
    free(H); free(tb);
    return res;
}

/* ---- experiment 0x80000: NGMLR's convex gap cost, exactly (Sedlazeck 2018, Methods; ngmlr 0.2.7 defaults, envs/telr.yml:48) ----
 * NGMLR charges a gap `open` once and every gap base an extension that DECAYS with the length the gap already has:
 * ext(i) = max(ext_min, ext_max - decay * i) for the base that makes a gap of length i one longer; pacbio: match 2, mismatch 5,
 * open 5, ext 5 -> 1, decay 0.15; ont: 1, 1, 1, 1 -> 0.5, 0.15.  NGMLR keeps the current gap length with every gap cell, so the
 * recurrence is the affine one with a length-dependent extension -- restated here in integers by scaling every score with
 * 20 (0.15 = 3/20): pacbio ext(i) = max(20, 100 - 3 i), ont ext(i) = max(10, 20 - 3 i).  The presets of the spec replace this
 * by the lower envelope of two affine pieces (exact at L = 1 and from L = 27 / 5 on); this variant measures what
 * that costs (tests/test_faithful_gate.py, tools/faithful_table.py).  Cells: H, E (gap in the query: target base consumed),
 * F, and the lengths LE, LF of the gaps E / F end with; ties as in band_dp (diagonal > E > F; "opened" before "extended"). */
static dp_res_t band_dp_convex(const dp_seq_t *s, int dlo, int dhi, int ext, const telr_map_opt *mo, const convex_t *C, u32v_t *rev_cig)
{
    const int m = s->m, n = s->n, D = dhi - dlo + 1, stride = (D + 2) / 2;
    dp_res_t res = { 0, 0, 0, 0, 0 };
    const int NEGX = -(1 << 29);
    int32_t *H = (int32_t*)malloc(4 * (size_t)(D + 2) * 5), *E = H + (D + 2), *F = E + (D + 2), *LE = F + (D + 2), *LF = LE + (D + 2);
    for (int x = 0; x < (D + 2) * 3; ++x) H[x] = NEGX;
    for (int x = 0; x < (D + 2) * 2; ++x) LE[x] = 0;
    uint8_t *tb = (uint8_t*)calloc((size_t)(m + n + 1) * stride, 1);
#define IX(d) ((d) - dlo + 1)
    int best = 0, bi = 0, bj = 0, prev_cur = NEGX;
    if (0 >= dlo && 0 <= dhi) H[IX(0)] = 0;
    for (int a = 1; a <= m + n; ++a) {
        int d0 = -a > dlo ? -a : dlo; if (a - 2 * m > d0) d0 = a - 2 * m;
        int d1 = a < dhi ? a : dhi;   if (2 * n - a < d1) d1 = 2 * n - a;
        if (((d0 - a) & 1) != 0) ++d0;
        int cur = NEGX, cur_d = 0;
        for (int d = d0; d <= d1; d += 2) {
            int i = (a - d) >> 1, j = (a + d) >> 1, x = IX(d);
            int32_t h, ve, vf, le = 0, lf = 0; uint8_t t = 0;
            if (i == 0) { ve = (int32_t)-cx_cost(C, j); le = j; vf = NEGX; h = ve; }
            else if (j == 0) { vf = (int32_t)-cx_cost(C, i); lf = i; ve = NEGX; h = vf; }
            else {
                int32_t hl = H[x - 1], hu = H[x + 1], hd = H[x], o, g;
                o = hl - C->open - cx_ext(C, 0); g = E[x - 1] - cx_ext(C, LE[x - 1]);
                if (g > o) { ve = g; le = LE[x - 1] + 1; t |= 8; } else { ve = o; le = 1; }
                o = hu - C->open - cx_ext(C, 0); g = F[x + 1] - cx_ext(C, LF[x + 1]);
                if (g > o) { vf = g; lf = LF[x + 1] + 1; t |= 16; } else { vf = o; lf = 1; }
                int qb = qbase(s, i - 1), tbv = tbase(s, j - 1);
                int sc = (qb > 3 || tbv > 3) ? -C->amb : (qb == tbv ? C->a : -C->b);
                h = hd + sc; int src = 0;
                if (ve > h) h = ve, src = 1;
                if (vf > h) h = vf, src = 2;
                t |= (uint8_t)src;
                tb[(size_t)a * stride + ((d - dlo) >> 1)] = t;
                ++res.cells;
            }
            if (h < NEGX) h = NEGX;
            if (ve < NEGX) ve = NEGX;
            if (vf < NEGX) vf = NEGX;
            if (le > C->flat) le = C->flat;          /* beyond it the extension is flat: the length need not grow */
            if (lf > C->flat) lf = C->flat;
            H[x] = h; E[x] = ve; F[x] = vf; LE[x] = le; LF[x] = lf;
            if (h > cur) cur = h, cur_d = d;
        }
        if (ext) {
            if (cur > best) best = cur, bi = (a - cur_d) >> 1, bj = (a + cur_d) >> 1;
            int c2 = cur > prev_cur ? cur : prev_cur;
            if (best - c2 > mo->zdrop * C->S) break;
            prev_cur = cur;
        }
    }
    int i, j;
    if (ext) { res.score = best; i = bi; j = bj; }
    else { res.score = H[IX(n - m)]; i = m; j = n; }
    res.bi = i; res.bj = j;
    int state = 0;
    while (i > 0 && j > 0) {
        uint8_t t = tb[(size_t)(i + j) * stride + ((j - i - dlo) >> 1)];
        if (j - i - dlo <= mo->fill_margin || dhi - (j - i) <= mo->fill_margin) res.touched = 1;
        if (state == 0) state = t & 7;
        if (state == 0) { cig_push(rev_cig, 0, 1); --i; --j; }
        else if (state == 1) { cig_push(rev_cig, 2, 1); if (!(t & 8))  state = 0; --j; }
        else                 { cig_push(rev_cig, 1, 1); if (!(t & 16)) state = 0; --i; }
    }
    if (i > 0) cig_push(rev_cig, 1, i);
    if (j > 0) cig_push(rev_cig, 2, j);
#undef IX
    free(H); free(tb);
    return res;
}

/* band wider than DP_DMAX: align min(m,n) bases on the main diagonal and close with one gap */
static dp_res_t band_dp_fallback(const dp_seq_t *s, const telr_map_opt *mo, u32v_t *rev_cig, int *mlen)
{
    dp_res_t r = { 0, s->m, s->n, 0, 0 };
    int mn = s->m < s->n ? s->m : s->n, g = s->m > s->n ? s->m - s->n : s->n - s->m;
    *mlen = 0;
    for (int x = 0; x < mn; ++x) {
        int qb = qbase(s, x), tbv = tbase(s, x);
        if (qb > 3 || tbv > 3) r.score -= mo->sc_ambi; else if (qb == tbv) { r.score += mo->a; ++*mlen; } else r.score -= mo->b;
    }
    if (g) { int c1 = mo->q + g * mo->e, c2 = mo->q2 + g * mo->e2; r.score -= c1 < c2 ? c1 : c2; cig_push(rev_cig, s->m > s->n ? 1 : 2, g); }
    cig_push(rev_cig, 0, mn);
    return r;
}

/* Adaptive band of a gap-fill segment.  First pass: W = 2 + fill_band_q4 * floor(sqrt(min(m,n))) / 16 diagonals either
 * side of the corner-to-corner diagonal range (indel drift between two anchors grows with the square root of the
 * distance); if the optimal path of that pass touches a band edge, the segment is re-aligned once with the wide band
 * (24 + mn/8, flatter above 512), itself limited so that the band stays within 1024 diagonals where the first-pass
 * width allows it.  Segments with m+n > ADAPT_MAX_STEPS use the wide band at once. */
static inline int even_lo(int lo) { return lo - (lo & 1); }
/* Long-gap fill (spec 3.11): a segment whose two lengths differ by more than bw (a link only the long join admits).  The
 * optimal path runs near the main diagonal from the start, jumps by ONE long gap, and runs near the main diagonal of the end:
 * LEFT = global banded DP from the start in the band |j - i| <= W over the first min(len, S + W) rows / columns (S = the
 * shorter length, W = ext_band), RIGHT = the same from the end on the reversed sequences; for an insertion-type segment
 * (m > n) the junction is the target column jL (and jR = n - jL) that maximises
 *     max_iL [HL(iL, jL) + e2 iL]  +  max_iR [HR(iR, jR) + e2 iR]        (smallest jL, then smallest iL / iR on ties)
 * i.e. the best total with a gap of m - iL - iR query bases charged on the second affine piece (the gap is longer than
 * bw - 2W, far beyond where the pieces cross); deletion-type segments swap the roles.  Cells, tie-breaks and trace-back
 * bytes are those of band_dp. */
typedef struct { int32_t *H; uint8_t *tb; int m, n, dlo, dhi, D, stride; int64_t cells; } band_all_t;
static void band_dp_all(const dp_seq_t *s, int W, const telr_map_opt *mo, band_all_t *B)
{
    const int m = s->m, n = s->n, dlo = even_lo(-W), dhi = W, D = dhi - dlo + 1, stride = (D + 2) / 2;
    const int q1 = mo->q, e1 = mo->e, q2 = mo->q2, e2 = mo->e2;
    int32_t *H = (int32_t*)malloc(4 * (size_t)(D + 2) * 5), *E1 = H + (D + 2), *F1 = E1 + (D + 2), *E2 = F1 + (D + 2), *F2 = E2 + (D + 2);
    for (int x = 0; x < (D + 2) * 5; ++x) H[x] = NEG;
    B->m = m; B->n = n; B->dlo = dlo; B->dhi = dhi; B->D = D; B->stride = stride; B->cells = 0;
    B->H = (int32_t*)malloc(4 * (size_t)(m + n + 1) * D);
    B->tb = (uint8_t*)calloc((size_t)(m + n + 1) * stride, 1);
    for (size_t x = 0; x < (size_t)(m + n + 1) * D; ++x) B->H[x] = NEG;
#define IX(d) ((d) - dlo + 1)
    H[IX(0)] = 0; B->H[0 * D + (0 - dlo)] = 0;
    for (int a = 1; a <= m + n; ++a) {
        int d0 = -a > dlo ? -a : dlo; if (a - 2 * m > d0) d0 = a - 2 * m;
        int d1 = a < dhi ? a : dhi;   if (2 * n - a < d1) d1 = 2 * n - a;
        if (((d0 - a) & 1) != 0) ++d0;
        for (int d = d0; d <= d1; d += 2) {
            int i = (a - d) >> 1, j = (a + d) >> 1, x = IX(d);
            int32_t h, ve1, vf1, ve2, vf2; uint8_t t = 0;
            if (i == 0) { ve1 = -(q1 + j * e1); ve2 = -(q2 + j * e2); vf1 = vf2 = NEG; h = ve1 > ve2 ? ve1 : ve2; }
            else if (j == 0) { vf1 = -(q1 + i * e1); vf2 = -(q2 + i * e2); ve1 = ve2 = NEG; h = vf1 > vf2 ? vf1 : vf2; }
            else {
                int32_t hl = H[x - 1], hu = H[x + 1], hd = H[x], o, g;
                o = hl - q1 - e1; g = E1[x - 1] - e1; if (g > o) ve1 = g, t |= 8;  else ve1 = o;
                o = hu - q1 - e1; g = F1[x + 1] - e1; if (g > o) vf1 = g, t |= 16; else vf1 = o;
                o = hl - q2 - e2; g = E2[x - 1] - e2; if (g > o) ve2 = g, t |= 32; else ve2 = o;
                o = hu - q2 - e2; g = F2[x + 1] - e2; if (g > o) vf2 = g, t |= 64; else vf2 = o;
                int qb = qbase(s, i - 1), tbv = tbase(s, j - 1);
                int sc = (qb > 3 || tbv > 3) ? -mo->sc_ambi : (qb == tbv ? mo->a : -mo->b);
                h = hd + sc; int src = 0;
                if (ve1 > h) h = ve1, src = 1;
                if (vf1 > h) h = vf1, src = 2;
                if (ve2 > h) h = ve2, src = 3;
                if (vf2 > h) h = vf2, src = 4;
                t |= (uint8_t)src;
                B->tb[(size_t)a * stride + ((d - dlo) >> 1)] = t;
                ++B->cells;
            }
            if (h < NEG) h = NEG;
            if (ve1 < NEG) ve1 = NEG;
            if (vf1 < NEG) vf1 = NEG;
            if (ve2 < NEG) ve2 = NEG;
            if (vf2 < NEG) vf2 = NEG;
            H[x] = h; E1[x] = ve1; F1[x] = vf1; E2[x] = ve2; F2[x] = vf2;
            B->H[(size_t)a * D + (d - dlo)] = h;
        }
    }
#undef IX
    free(H);
}
/* walk from cell (i, j) of a band_dp_all matrix to (0, 0): ops in end -> start order, one per column */
static void band_all_walk(const band_all_t *B, int i, int j, u32v_t *out)
{
    int state = 0;
    while (i > 0 && j > 0) {
        uint8_t t = B->tb[(size_t)(i + j) * B->stride + ((j - i - B->dlo) >> 1)];
        if (state == 0) state = t & 7;
        if (state == 0) { vpush(uint32_t, *out, 0); --i; --j; }
        else if (state == 1) { vpush(uint32_t, *out, 2); if (!(t & 8))  state = 0; --j; }
        else if (state == 2) { vpush(uint32_t, *out, 1); if (!(t & 16)) state = 0; --i; }
        else if (state == 3) { vpush(uint32_t, *out, 2); if (!(t & 32)) state = 0; --j; }
        else                 { vpush(uint32_t, *out, 1); if (!(t & 64)) state = 0; --i; }
    }
    for (; i > 0; --i) vpush(uint32_t, *out, 1);
    for (; j > 0; --j) vpush(uint32_t, *out, 2);
}
static dp_res_t longgap_fill(const dp_seq_t *s, const telr_map_opt *mo, u32v_t *rev_cig)
{
    const int m = s->m, n = s->n, ins = m > n, S = ins ? n : m, W = mo->ext_band;
    const int lm = m < S + W ? m : S + W, ln = n < S + W ? n : S + W;
    dp_seq_t sl = *s, sr = *s;
    sl.m = lm; sl.n = ln;
    sr.m = lm; sr.n = ln; sr.qstep = -s->qstep; sr.tstep = -s->tstep;
    sr.qi0 = s->qi0 + (int64_t)s->qstep * (m - 1); sr.ti0 = s->ti0 + (int64_t)s->tstep * (n - 1);
    band_all_t L, R;
    band_dp_all(&sl, W, mo, &L); band_dp_all(&sr, W, mo, &R);
    const int e2 = mo->e2;
    /* best junction */
    int64_t best = INT64_MIN; int bjl = 0, bil = 0, bir = 0;
    for (int c = 0; c <= S; ++c) {                 /* c: short-axis coordinate consumed by LEFT (target column for ins, query row for del) */
        int64_t vl = INT64_MIN, vr = INT64_MIN; int al = 0, ar = 0;
        for (int k = c - W; k <= c + W; ++k) {     /* long-axis coordinate inside the band */
            if (k < 0 || k > (ins ? lm : ln)) continue;
            int i = ins ? k : c, j = ins ? c : k, d = j - i;
            if (d < L.dlo || d > L.dhi) continue;
            int32_t h = L.H[(size_t)(i + j) * L.D + (d - L.dlo)];
            if (h <= NEG / 2) continue;
            int64_t v = (int64_t)h + (int64_t)e2 * k;
            if (v > vl) vl = v, al = k;
        }
        const int c2 = S - c;
        for (int k = c2 - W; k <= c2 + W; ++k) {
            if (k < 0 || k > (ins ? lm : ln)) continue;
            int i = ins ? k : c2, j = ins ? c2 : k, d = j - i;
            if (d < R.dlo || d > R.dhi) continue;
            int32_t h = R.H[(size_t)(i + j) * R.D + (d - R.dlo)];
            if (h <= NEG / 2) continue;
            int64_t v = (int64_t)h + (int64_t)e2 * k;
            if (v > vr) vr = v, ar = k;
        }
        if (vl == INT64_MIN || vr == INT64_MIN) continue;
        if ((ins ? m : n) - al - ar < 1) continue;
        if (vl + vr > best) best = vl + vr, bjl = c, bil = al, bir = ar;
    }
    dp_res_t r = { 0, m, n, 0, 0 };
    r.cells = (int32_t)(L.cells + R.cells);
    /* (S = 0 cannot fail: c = 0 with the corner cells always exists) */
    const int g = (ins ? m : n) - bil - bir;
    const int li = ins ? bil : bjl, lj = ins ? bjl : bil, ri = ins ? bir : S - bjl, rj = ins ? S - bjl : bir;
    const int c1 = mo->q + g * mo->e, c2g = mo->q2 + g * mo->e2;
    r.score = L.H[(size_t)(li + lj) * L.D + (lj - li - L.dlo)] + R.H[(size_t)(ri + rj) * R.D + (rj - ri - R.dlo)] - (c1 < c2g ? c1 : c2g);
    /* ops: rev_cig holds end -> start.  RIGHT's walk runs junction -> end in the original orientation, so it is pushed reversed */
    u32v_t wl = {0, 0, 0}, wr = {0, 0, 0};
    band_all_walk(&L, li, lj, &wl); band_all_walk(&R, ri, rj, &wr);
    for (int64_t z = wr.n - 1; z >= 0; --z) cig_push(rev_cig, (int)wr.a[z], 1);
    cig_push(rev_cig, ins ? 1 : 2, g);
    for (int64_t z = 0; z < wl.n; ++z) cig_push(rev_cig, (int)wl.a[z], 1);
    free(wl.a); free(wr.a); free(L.H); free(L.tb); free(R.H); free(R.tb);
    return r;
}

static inline int isqrt32(int v)
{
    int r = 0;
    for (int b = 1 << 15; b; b >>= 1) { int t = r | b; if ((int64_t)t * t <= v) r = t; }
    return r;
}
static inline int fill_band(int m, int n, const telr_map_opt *mo)
{
    int mn = m < n ? m : n, q4 = mo->fill_band_q4 > 0 ? mo->fill_band_q4 : 8;
    int W = 2 + ((q4 * isqrt32(mn)) >> 4);
    return W < mo->bw ? W : mo->bw;
}
#define ADAPT_MAX_STEPS 1000      /* m+n above which a segment skips the narrow pass */
static inline int fill_band_wide(int m, int n, const telr_map_opt *mo)
{
    int mn = m < n ? m : n, dl = n - m, adl = dl < 0 ? -dl : dl;
    int W = mn <= 512 ? 24 + (mn >> 3) : 88 + ((mn - 512) >> 4);
    if (W > mo->bw) W = mo->bw;
    int cap = (1022 - adl) / 2, Wn = fill_band(m, n, mo);
    if (cap < Wn) cap = Wn;
    return W < cap ? W : cap;
}
/* the lower band edge is rounded down to an even diagonal (the GPU pairs diagonals per lane) */

/* debug entry: one global banded alignment of two ASCII strings */
int32_t tor_nw(const char *q, int m, const char *t, int n, const telr_map_opt *mo, uint32_t *cig, int32_t *n_cig, int32_t cap)
{
    uint8_t *qq = (uint8_t*)malloc(m + 1), *tt = (uint8_t*)malloc(n + 1);
    for (int i = 0; i < m; ++i) qq[i] = NT4[(uint8_t)q[i]];
    for (int i = 0; i < n; ++i) tt[i] = NT4[(uint8_t)t[i]];
    dp_seq_t s = { qq, tt, 0, 0, 1, 1, 0, m, n };
    int W = fill_band(m, n, mo), dl = n - m;
    u32v_t rc = {0, 0, 0};
    dp_res_t r = band_dp(&s, even_lo((dl < 0 ? dl : 0) - W), (dl > 0 ? dl : 0) + W, 0, mo, &rc);
    *n_cig = (int32_t)rc.n;
    for (int64_t i = 0; i < rc.n && i < cap; ++i) cig[i] = rc.a[rc.n - 1 - i];
    free(rc.a); free(qq); free(tt);
    return r.score;
}

/* debug entry: one z-drop extension */
int32_t tor_ext(const char *q, int m, const char *t, int n, const telr_map_opt *mo, uint32_t *cig, int32_t *n_cig, int32_t cap,
                int32_t *qend, int32_t *tend)
{
    uint8_t *qq = (uint8_t*)malloc(m + 1), *tt = (uint8_t*)malloc(n + 1);
    for (int i = 0; i < m; ++i) qq[i] = NT4[(uint8_t)q[i]];
    for (int i = 0; i < n; ++i) tt[i] = NT4[(uint8_t)t[i]];
    int mq = m < mo->ext_max ? m : mo->ext_max, mt = n < mq + mo->ext_band ? n : mq + mo->ext_band;
    dp_seq_t s = { qq, tt, 0, 0, 1, 1, 0, mq, mt };
    u32v_t rc = {0, 0, 0};
    dp_res_t r = band_dp(&s, even_lo(-mo->ext_band), mo->ext_band, 1, mo, &rc);
    *n_cig = (int32_t)rc.n; *qend = r.bi; *tend = r.bj;
    for (int64_t i = 0; i < rc.n && i < cap; ++i) cig[i] = rc.a[rc.n - 1 - i];
    free(rc.a); free(qq); free(tt);
    return r.score;
}

/* align one chain; fills the alignment fields of `al` and appends the CIGAR */
static void align_chain(const tor_index *ix, const uint8_t *q, int qlen, const chain_t *c, const uint64_t *ca,
                        const telr_map_opt *mo, telr_aln *al, u32v_t *cigars, telr_counters *ctr)
{
    const uint8_t *t = ix->seq[c->tid];
    const int tlen = ix->len[c->tid], go = (int32_t)ix->goff[c->tid];
    /* TELR_MF_FAITHFUL (oracle only): no speed-motivated bounds -- gap fills over the whole -r band, end extensions over
     * the whole remaining read in a band of -r diagonals (still z-drop terminated), chaining look-back 5000.  The drift of
     * the tuned presets against this mode is gated by tests/test_faithful_gate.py. */
    const int faithful = (mo->flags & (TELR_MF_FAITHFUL | 0x200)) != 0, faithful_ext = (mo->flags & (TELR_MF_FAITHFUL | 0x400)) != 0;
    const int ext_max = faithful_ext ? (1 << 30) : mo->ext_max, ext_band = faithful_ext ? mo->bw : mo->ext_band;
    convex_t CX; const int cx = (mo->cx_scale > 0 || (mo->flags & MFX_CONVEX)) && convex_of(mo, &CX);       /* segment scores come back in 1/S units */
    /* query accessor on the chain's strand */
    dp_seq_t s; s.q = q; s.t = t; s.qcomp = c->rev;
    /* breakpoints */
    VEC(int32_t) bp = {0, 0, 0};
    int32_t r0 = c->rs, q0 = c->qs;
    vpush(int32_t, bp, r0); vpush(int32_t, bp, q0);
    int32_t lr = r0, lq = q0;
    for (int32_t i = 0; i < c->cnt; ++i) {
        int32_t cr = A_G(ca[i]) - go + 1, cq = A_Q(ca[i]) + 1;
        if (i == c->cnt - 1 || (cq - lq >= mo->min_ksw_len && cr - lr >= mo->min_ksw_len)) {
            vpush(int32_t, bp, cr); vpush(int32_t, bp, cq); lr = cr; lq = cq;
        }
    }
    int nseg = (int)(bp.n / 2) - 1;
    u32v_t cig = {0, 0, 0}, rc = {0, 0, 0};
    int32_t dp = 0, n_zdrop = 0;
    /* left extension: reversed sequences starting at (q0-1, r0-1) going down */
    int32_t qs = q0, rs = r0;
    if (q0 > 0 && r0 > 0) {
        int mq = q0 < ext_max ? q0 : ext_max, mt = r0 < mq + ext_band ? r0 : mq + ext_band;
        s.m = mq; s.n = mt; s.tstep = -1; s.ti0 = r0 - 1;
        if (c->rev) { s.qstep = 1; s.qi0 = qlen - q0; } else { s.qstep = -1; s.qi0 = q0 - 1; }
        rc.n = 0;
        dp_res_t r = band_dp(&s, even_lo(-ext_band), ext_band, 1, mo, &rc);
        ++ctr->dp_problems; ctr->dp_cells += r.cells; ctr->window_bases += mt;
        dp += r.score; qs = q0 - r.bi; rs = r0 - r.bj;
        /* rev_cig is end->start of the reversed problem == left-to-right on the forward sequences */
        for (int64_t z = 0; z < rc.n; ++z) cig_push(&cig, rc.a[z] & 0xf, rc.a[z] >> 4);
    }
    for (int g = 0; g < nseg; ++g) {
        int32_t sr = bp.a[2 * g], sq = bp.a[2 * g + 1], er = bp.a[2 * g + 2], eq = bp.a[2 * g + 3];
        s.m = eq - sq; s.n = er - sr; s.tstep = 1; s.ti0 = sr;
        if (c->rev) { s.qstep = -1; s.qi0 = qlen - 1 - sq; } else { s.qstep = 1; s.qi0 = sq; }
        /* long segments are few and a second pass over one of them is slow: they take the wide band at once */
        const int is_long = s.m + s.n > ADAPT_MAX_STEPS;
        int W = is_long ? fill_band_wide(s.m, s.n, mo) : fill_band(s.m, s.n, mo), dl = s.n - s.m;
        if (faithful) W = mo->bw;               /* the whole -r band, whatever the segment */
        rc.n = 0;
        int lo = even_lo((dl < 0 ? dl : 0) - W), hi = (dl > 0 ? dl : 0) + W, fb_mlen;
        if (faithful) { if (lo < -s.m) lo = even_lo(-s.m); if (hi > s.n) hi = s.n; }      /* no wider than the matrix */
        const int longgap = mo->bw_long > mo->bw && (dl > mo->bw || -dl > mo->bw);
        dp_res_t r = longgap ? longgap_fill(&s, mo, &rc) : (!faithful && hi - lo + 1 > DP_DMAX) ? band_dp_fallback(&s, mo, &rc, &fb_mlen) : band_dp(&s, lo, hi, 0, mo, &rc);
        if (cx && (longgap || (!faithful && hi - lo + 1 > DP_DMAX))) {
            /* the diagonal fall-back closes with ONE gap: its convex cost; (a long-gap fill keeps the envelope: telr_map refuses cx_scale with bw_long) */
            if (!longgap) { int g = s.m > s.n ? s.m - s.n : s.n - s.m, c1 = mo->q + g * mo->e, c2 = mo->q2 + g * mo->e2; r.score = (r.score + (g ? (c1 < c2 ? c1 : c2) : 0)) * CX.S - (g ? (int)cx_cost(&CX, g) : 0); }
            else r.score *= CX.S;
        }
        ++ctr->dp_problems; ctr->dp_cells += r.cells; ctr->window_bases += s.n;
        if (r.touched && !is_long && !faithful && !longgap) {            /* second pass with the wide band */
            int W2 = fill_band_wide(s.m, s.n, mo);
            lo = even_lo((dl < 0 ? dl : 0) - W2); hi = (dl > 0 ? dl : 0) + W2;
            if (W2 > W && hi - lo + 1 <= DP_DMAX) { rc.n = 0; r = band_dp(&s, lo, hi, 0, mo, &rc); ctr->dp_cells += r.cells; }
        }
        dp += r.score;
        if (mo->flags & MFX_ZSPLIT) {
            /* the running score along the fill's path against its running maximum (ksw2's z-drop test with the diagonal term) */
            int32_t sc = 0, mx = 0, mi = 0, mj = 0, i_ = 0, j_ = 0, dropped = 0;
            for (int64_t z = rc.n - 1; z >= 0 && !dropped; --z) {
                int op = rc.a[z] & 0xf, ln = rc.a[z] >> 4;
                if (op == 0) {
                    for (int x = 0; x < ln && !dropped; ++x) {
                        int qb = qbase(&s, i_), tb = tbase(&s, j_);
                        sc += (qb < 4 && tb < 4) ? (qb == tb ? mo->a : -mo->b) : -mo->sc_ambi;
                        ++i_; ++j_;
                        if (sc > mx) { mx = sc; mi = i_; mj = j_; }
                        else { int dd = (i_ - mi) - (j_ - mj); if (dd < 0) dd = -dd; if (mx - sc > mo->zdrop + mo->e2 * dd) dropped = 1; }
                    }
                } else {
                    int c1 = mo->q + mo->e * ln, c2 = mo->q2 + mo->e2 * ln;
                    sc -= c1 < c2 ? c1 : c2;
                    if (op == 1) i_ += ln; else j_ += ln;
                    int dd = (i_ - mi) - (j_ - mj); if (dd < 0) dd = -dd;
                    if (mx - sc > mo->zdrop + mo->e2 * dd) dropped = 1;
                }
            }
            n_zdrop += dropped;
        }
        for (int64_t z = rc.n - 1; z >= 0; --z) cig_push(&cig, rc.a[z] & 0xf, rc.a[z] >> 4);
    }
    int32_t qe = c->qe, re = c->re;
    if (qe < qlen && re < tlen) {
        int rq = qlen - qe, rt = tlen - re;
        int mq = rq < ext_max ? rq : ext_max, mt = rt < mq + ext_band ? rt : mq + ext_band;
        s.m = mq; s.n = mt; s.tstep = 1; s.ti0 = re;
        if (c->rev) { s.qstep = -1; s.qi0 = qlen - 1 - qe; } else { s.qstep = 1; s.qi0 = qe; }
        rc.n = 0;
        dp_res_t r = band_dp(&s, even_lo(-ext_band), ext_band, 1, mo, &rc);
        ++ctr->dp_problems; ctr->dp_cells += r.cells; ctr->window_bases += mt;
        dp += r.score; qe += r.bi; re += r.bj;
        for (int64_t z = rc.n - 1; z >= 0; --z) cig_push(&cig, rc.a[z] & 0xf, rc.a[z] >> 4);
    }
    /* statistics from the final CIGAR */
    int32_t mlen = 0, blen = 0, nambi = 0, qi = qs, ti = rs;
    for (int64_t z = 0; z < cig.n; ++z) {
        int op = cig.a[z] & 0xf, len = cig.a[z] >> 4;
        blen += len;
        if (op == 0) {
            for (int x = 0; x < len; ++x) {
                int qb = c->rev ? q[qlen - 1 - (qi + x)] : q[qi + x];
                if (c->rev && qb < 4) qb = 3 - qb;
                int tbv = t[ti + x];
                if (qb < 4 && qb == tbv) ++mlen;
            }
            qi += len; ti += len;
        } else if (op == 1) qi += len; else ti += len;
    }
    al->ts = rs; al->te = re;
    if (c->rev) { al->qs = qlen - qe; al->qe = qlen - qs; } else { al->qs = qs; al->qe = qe; }
    if (cx) { int32_t v = dp + CX.S / 2; dp = v >= 0 ? v / CX.S : -((-v + CX.S - 1) / CX.S); }      /* back to the preset's units: floor((sum + S/2) / S) */
    al->mlen = mlen; al->blen = blen; al->n_ambi = n_zdrop; al->dp_score = dp; (void)nambi;      /* n_ambi: 0 unless the 0x10000 experiment counts z-dropped fills */
    al->n_cigar = (int32_t)cig.n; al->cigar_off = cigars->n;
    for (int64_t z = 0; z < cig.n; ++z) vpush(uint32_t, *cigars, cig.a[z]);
    free(cig.a); free(rc.a); free(bp.a);
}

/* ------------------------------------------------------------------------- */
/* 7. the mapping driver                                                      */
typedef struct tor_result {
    VEC(telr_aln) alns;
    u32v_t cigars;
    telr_counters ctr;
    /* debug captures (concatenated over queries) */
    int debug;
    u64v_t d_anchor; VEC(int64_t) d_anchor_off; VEC(int32_t) d_f, d_p;
    VEC(int32_t) d_chain;  /* per chain: qid, score, cnt, rev, tid, rs, re, qs, qe */
} tor_result;

static int32_t mapq_of(const telr_aln *r, const telr_map_opt *mo)
{
    if (!(r->flags & TELR_F_PRIMARY) && !(r->flags & TELR_F_SUPPL)) return 0;
    if (mo->flags & MFX_MAPQ) {
        /* mm_set_mapq of minimap2 2.22 without the second-best DP score (ksw2's dp_max2 is not computed here): identity,
         * chain-score and anchor-count penalties, the sub-optimal chain, and the log of the number of sub-optimal chains */
        if (!(r->flags & TELR_F_PRIMARY) && !(r->flags & TELR_F_SUPPL)) return 0;
        float pen_s1 = r->score > 100 ? 1.0f : 0.01f * (float)r->score;
        float pen_cm = r->cnt > 10 ? 1.0f : 0.1f * (float)r->cnt;
        if (pen_s1 < pen_cm) pen_cm = pen_s1;
        float subsc = (float)(r->subsc > mo->min_chain_score ? r->subsc : mo->min_chain_score);
        float identity = r->blen > 0 ? (float)r->mlen / (float)r->blen : 0.0f;
        float x = subsc / (float)r->score;
        int32_t mq = (int32_t)(identity * pen_cm * 40.0f * (1.0f - x) * logf((float)r->dp_score / (float)mo->a));
        mq -= (int32_t)(4.343f * logf((float)r->n_sub + 1.0f) + .499f);
        if (mq < 0) mq = 0;
        if (mq > 60) mq = 60;
        return mq;
    }
    float f1 = (float)r->score, f2 = (float)(r->subsc > mo->min_chain_score ? r->subsc : mo->min_chain_score);
    float pen_cm = r->cnt > 10 ? 1.0f : 0.1f * (float)r->cnt;
    float x = f2 / f1; if (x > 1.0f) x = 1.0f;
    int32_t mq = (int32_t)(40.0f * (1.0f - x) * pen_cm * logf(f1));
    if (mq > 60) mq = 60;
    if (mq < 0) mq = 0;
    return mq;
}

tor_result *tor_map(const tor_index *ix, int32_t nq, const char *ascii, const int64_t *off, const int32_t *len,
                    const int32_t *qtarget, const telr_map_opt *mo, int debug)
{
    tor_result *R = (tor_result*)calloc(1, sizeof(*R));
    R->debug = debug;
    int32_t mid_occ = tor_mid_occ(ix, mo->mid_occ_frac, mo->min_mid_occ, mo->max_mid_occ);
    for (int32_t qi = 0; qi < nq; ++qi) {
        int qlen = len[qi];
        uint8_t *q = (uint8_t*)malloc(qlen > 0 ? qlen : 1);
        for (int i = 0; i < qlen; ++i) q[i] = NT4[(uint8_t)ascii[off[qi] + i]];
        R->ctr.query_bases += qlen;
        u64v_t an = {0, 0, 0};
        collect_anchors(ix, q, qlen, qtarget ? qtarget[qi] : -1, qtarget == NULL, mid_occ, mo, &an, &R->ctr.minimizers, &R->ctr.probes);
        R->ctr.anchors += an.n;
        int32_t *f = (int32_t*)malloc(4 * (an.n ? an.n : 1)), *p = (int32_t*)malloc(4 * (an.n ? an.n : 1));
        telr_map_opt mc = *mo;
        if (mo->flags & MFX_LONGJOIN) mc.bw_long = 0;                           /* first round of the experiment: the short bandwidth only */
        if ((mo->flags & MFX_RMQ) && mo->bw >= 10000) mc.bw = 100000;          /* asm10 of minimap2 2.22: --rmq -r100k -g10k */
        chain_dp(an.a, an.n, &mc, f, p);
        chainv_t ch = {0, 0, 0}; u64v_t ca = {0, 0, 0};
        chain_backtrack(ix, an.a, an.n, f, p, &mc, &ch, &ca);
        if ((mo->flags & MFX_LONGJOIN) && ch.n > 1 && mc.bw < 20000) {          /* minimap2's two rounds: re-chain with bw_long only when the first round left several chains */
            mc.bw_long = 20000; if (!(mo->flags & 0x40000)) mc.flags |= MFX_LOOKBACK;      /* 0x40000: keep the preset's look-back in the second round */
            ch.n = 0; ca.n = 0;
            chain_dp(an.a, an.n, &mc, f, p);
            chain_backtrack(ix, an.a, an.n, f, p, &mc, &ch, &ca);
        }
        R->ctr.chains += ch.n;
        if (debug) {
            vpush(int64_t, R->d_anchor_off, R->d_anchor.n);
            for (int64_t i = 0; i < an.n; ++i) { vpush(uint64_t, R->d_anchor, an.a[i]); vpush(int32_t, R->d_f, f[i]); vpush(int32_t, R->d_p, p[i]); }
            for (int64_t i = 0; i < ch.n; ++i) {
                chain_t *c = &ch.a[i];
                int32_t v[9] = { qi, c->score, c->cnt, c->rev, c->tid, c->rs, c->re, c->qs, c->qe };
                for (int z = 0; z < 9; ++z) vpush(int32_t, R->d_chain, v[z]);
            }
        }
        /* pass 1: selection on chain scores */
        int n = (int)ch.n;
        sel_t *s = (sel_t*)malloc(sizeof(sel_t) * (n ? n : 1));
        int32_t *cscore = (int32_t*)malloc(4 * (n ? n : 1));
        for (int i = 0; i < n; ++i) {
            chain_t *c = &ch.a[i];
            s[i].ci = i; s[i].key = c->score; s[i].ord = c->disc; s[i].tid = c->tid;
            if (c->rev) { s[i].fs = qlen - c->qe; s[i].fe = qlen - c->qs; } else { s[i].fs = c->qs; s[i].fe = c->qe; }
            cscore[i] = c->score;
        }
        qsort(s, n, sizeof(sel_t), cmp_sel);
        select_chains(s, n, mo, cscore);
        /* base-level alignment of the kept chains */
        telr_aln *al = (telr_aln*)calloc(n ? n : 1, sizeof(telr_aln));
        sel_t *s2 = (sel_t*)malloc(sizeof(sel_t) * (n ? n : 1));
        int n2 = 0;
        for (int i = 0; i < n; ++i) {
            if (!s[i].keep) continue;
            chain_t *c = &ch.a[s[i].ci];
            telr_aln *r = &al[s[i].ci];
            r->qid = qi; r->tid = c->tid; r->qlen = qlen; r->tlen = ix->len[c->tid];
            r->score = c->score; r->cnt = c->cnt; r->flags = c->rev ? TELR_F_REV : 0;
            if (mo->flags & TELR_MF_CIGAR) {
                align_chain(ix, q, qlen, c, ca.a + c->a_off, mo, r, &R->cigars, &R->ctr);
                if (r->dp_score < mo->min_dp_max) continue;
            } else {
                r->ts = c->rs; r->te = c->re;
                if (c->rev) { r->qs = qlen - c->qe; r->qe = qlen - c->qs; } else { r->qs = c->qs; r->qe = c->qe; }
                r->mlen = c->score < (c->qe - c->qs) ? c->score : (c->qe - c->qs);
                r->blen = (c->qe - c->qs) > (c->re - c->rs) ? (c->qe - c->qs) : (c->re - c->rs);
                r->dp_score = c->score;
            }
            s2[n2].ci = s[i].ci; s2[n2].key = r->dp_score; s2[n2].ord = n2; s2[n2].tid = c->tid;
            s2[n2].fs = r->qs; s2[n2].fe = r->qe;
            ++n2;
        }
        /* pass 2: selection on DP scores, then flags and mapq */
        qsort(s2, n2, sizeof(sel_t), cmp_sel);
        select_chains(s2, n2, mo, cscore);
        int64_t base = R->alns.n;
        int32_t *newidx = (int32_t*)malloc(4 * (n2 ? n2 : 1));
        int nk = 0;
        for (int i = 0; i < n2; ++i) newidx[i] = s2[i].keep ? nk++ : -1;
        int per_t = (mo->flags & TELR_MF_PER_TARGET) != 0;
        for (int i = 0; i < n2; ++i) {
            if (!s2[i].keep) continue;
            telr_aln r = al[s2[i].ci];
            r.parent = newidx[s2[i].parent]; r.subsc = s2[i].subsc; r.n_sub = s2[i].n_sub;
            if (s2[i].parent == i) {
                int first = 1;
                for (int j = 0; j < i; ++j) if (s2[j].keep && s2[j].parent == j && (!per_t || s2[j].tid == s2[i].tid)) { first = 0; break; }
                r.flags |= first ? TELR_F_PRIMARY : TELR_F_SUPPL;
            } else r.flags |= TELR_F_SECONDARY;
            r.mapq = mapq_of(&r, mo);
            R->ctr.cigar_ops += r.n_cigar;
            vpush(telr_aln, R->alns, r);
        }
        (void)base;
        R->ctr.records += nk;
        free(newidx); free(s2); free(al); free(cscore); free(s); free(ch.a); free(ca.a); free(f); free(p); free(an.a); free(q);
    }
    if (debug) vpush(int64_t, R->d_anchor_off, R->d_anchor.n);
    return R;
}

int64_t tor_result_count(const tor_result *r) { return r->alns.n; }
const telr_aln *tor_result_alns(const tor_result *r) { return r->alns.a; }
int64_t tor_result_cigar_count(const tor_result *r) { return r->cigars.n; }
const uint32_t *tor_result_cigars(const tor_result *r) { return r->cigars.a; }
void tor_result_counters(const tor_result *r, telr_counters *c) { *c = r->ctr; }
int64_t tor_debug_n_anchor(const tor_result *r) { return r->d_anchor.n; }
const uint64_t *tor_debug_anchors(const tor_result *r) { return r->d_anchor.a; }
const int64_t *tor_debug_anchor_off(const tor_result *r) { return r->d_anchor_off.a; }
const int32_t *tor_debug_f(const tor_result *r) { return r->d_f.a; }
const int32_t *tor_debug_p(const tor_result *r) { return r->d_p.a; }
int64_t tor_debug_n_chain(const tor_result *r) { return r->d_chain.n / 9; }
const int32_t *tor_debug_chains(const tor_result *r) { return r->d_chain.a; }
void tor_result_free(tor_result *r)
{
    if (!r) return;
    free(r->alns.a); free(r->cigars.a); free(r->d_anchor.a); free(r->d_anchor_off.a); free(r->d_f.a); free(r->d_p.a); free(r->d_chain.a); free(r);
}

/* ------------------------------------------------------------------------- */
/* 8. depth medians (samtools depth -aa -r | statistics.median restated;       */
/*    call site src/telr/TELR_te.py:870-884).  Counts M columns of records    */
/*    that are not secondary; deletions and insertions do not count.           */
void tor_depth_medians(const telr_aln *alns, int64_t n_aln, const uint32_t *cigars, int32_t n_targets, const int32_t *tlen,
                       int32_t n_iv, const int32_t *iv_tid, const int32_t *iv_s, const int32_t *iv_e, double *out)
{
    int32_t **depth = (int32_t**)calloc(n_targets, sizeof(int32_t*));
    for (int64_t i = 0; i < n_aln; ++i) {
        const telr_aln *r = &alns[i];
        if (r->flags & TELR_F_SECONDARY) continue;
        if (!depth[r->tid]) depth[r->tid] = (int32_t*)calloc(tlen[r->tid] + 1, 4);
        int32_t t = r->ts;
        for (int32_t z = 0; z < r->n_cigar; ++z) {
            uint32_t c = cigars[r->cigar_off + z]; int op = c & 0xf, l = c >> 4;
            if (op == 0) { for (int x = 0; x < l; ++x) ++depth[r->tid][t + x]; t += l; }
            else if (op == 2) t += l;
        }
    }
    for (int32_t v = 0; v < n_iv; ++v) {
        int32_t s = iv_s[v], e = iv_e[v], tid = iv_tid[v], L = tlen[tid];
        /* -aa prints every position of the region that exists on the target */
        if (s < 0) s = 0;
        if (e > L - 1) e = L - 1;
        int32_t n = e - s + 1;
        if (n <= 0) { out[v] = NAN; continue; }
        int32_t *a = (int32_t*)malloc(4 * n);
        for (int32_t x = 0; x < n; ++x) { a[x] = depth[tid] ? depth[tid][s + x] : 0; if (a[x] > DEPTH_CAP) a[x] = DEPTH_CAP; }
        qsort(a, n, 4, cmp_u32);
        out[v] = (n & 1) ? (double)a[n / 2] : ((double)a[n / 2 - 1] + (double)a[n / 2]) / 2.0;
        free(a);
    }
    for (int32_t i = 0; i < n_targets; ++i) free(depth[i]);
    free(depth);
}

/* ------------------------------------------------------------------------- */
/* 9. pile-up consensus (spec 3.12; f4: the polishing hand-off H3).  Majority vote over the PRIMARY records of a result   */
/*    (`samtools view -F0x900`, TELR_assembly.py:226-236) -- NOT wtpoa-cns's partial-order alignment.                      */
/*    Per target position p: base[b] = records with an M column at p whose query base is b, del = records with a D column,  */
/*    nq = M columns with an ambiguous query base, cov = sum of all; ins[k][b] / insn[k] = records with an inserted base    */
/*    number k (< CONS_KMAX) right after p.  Call: cov < min_depth -> the draft base; else drop p iff 2 del > cov, else the  */
/*    most frequent base (ties: the draft base if it is among them, else the smallest code; no vote at all: the draft);     */
/*    then inserted columns k = 0, 1, .. while 2 insn[k] > cov (most frequent base, ties smallest, none -> N).              */
/*    D runs longer than CONS_MAXDEL and inserted bases beyond CONS_KMAX do not vote: structural differences stay.          */
#define CONS_KMAX 8
#define CONS_MAXDEL 30     /* a longer D is a structural difference (a read of the other allele), not an error of the draft: it does not vote */
typedef struct { uint32_t b[4], del, nq, insn[CONS_KMAX], insb[CONS_KMAX][4]; } cons_cell_t;
/* q: the query sequences as nt4 codes; qoff/qlen per query; t: targets as ASCII (draft bases are copied as they are) */
int64_t tor_consensus(const telr_aln *alns, int64_t n_aln, const uint32_t *cigars, const uint8_t *q_nt4, const int64_t *qoff,
                      int32_t n_targets, const char *t_ascii, const int64_t *toff, const int32_t *tlen, int32_t min_depth,
                      char *out, int64_t cap, int64_t *out_off, int32_t *out_len)
{
    int64_t total = 0, w = 0;
    for (int32_t t = 0; t < n_targets; ++t) total += tlen[t];
    cons_cell_t *cell = (cons_cell_t*)calloc(total ? total : 1, sizeof(cons_cell_t));
    int64_t *base = (int64_t*)malloc(8 * (n_targets + 1));
    base[0] = 0; for (int32_t t = 0; t < n_targets; ++t) base[t + 1] = base[t] + tlen[t];
    for (int64_t i = 0; i < n_aln; ++i) {
        const telr_aln *r = &alns[i];
        if (r->flags & (TELR_F_SECONDARY | TELR_F_SUPPL)) continue;
        const int rev = (r->flags & TELR_F_REV) != 0;
        const uint8_t *q = q_nt4 + qoff[r->qid];
        int32_t qi = rev ? r->qlen - r->qe : r->qs, ti = r->ts;
        for (int32_t z = 0; z < r->n_cigar; ++z) {
            uint32_t c = cigars[r->cigar_off + z]; int op = c & 0xf, l = c >> 4;
            if (op == 0) {
                for (int x = 0; x < l; ++x) {
                    int qb = rev ? q[r->qlen - 1 - (qi + x)] : q[qi + x];
                    if (rev && qb < 4) qb = 3 - qb;
                    cons_cell_t *cc = &cell[base[r->tid] + ti + x];
                    if (qb < 4) ++cc->b[qb]; else ++cc->nq;
                }
                qi += l; ti += l;
            } else if (op == 2) { if (l <= CONS_MAXDEL) for (int x = 0; x < l; ++x) ++cell[base[r->tid] + ti + x].del; ti += l; }
            else {
                if (ti > r->ts) {          /* an insertion before the first aligned target base has no position to hang on */
                    cons_cell_t *cc = &cell[base[r->tid] + ti - 1];
                    for (int x = 0; x < l && x < CONS_KMAX; ++x) {
                        int qb = rev ? q[r->qlen - 1 - (qi + x)] : q[qi + x];
                        if (rev && qb < 4) qb = 3 - qb;
                        ++cc->insn[x];
                        if (qb < 4) ++cc->insb[x][qb];
                    }
                }
                qi += l;
            }
        }
    }
    for (int32_t t = 0; t < n_targets; ++t) {
        out_off[t] = w;
        for (int32_t p = 0; p < tlen[t]; ++p) {
            const cons_cell_t *cc = &cell[base[t] + p];
            const char draft = "ACGTN"[NT4[(uint8_t)t_ascii[toff[t] + p]]];        /* bases as the engine sees them */
            const uint32_t cov = cc->b[0] + cc->b[1] + cc->b[2] + cc->b[3] + cc->del + cc->nq;
            if ((int64_t)cov < min_depth) { if (w < cap) { out[w] = draft; } ++w; continue; }
            if (2 * cc->del <= cov) {
                int dcode = NT4[(uint8_t)draft], best = -1; uint32_t bv = 0;
                for (int b = 0; b < 4; ++b) if (cc->b[b] > bv) bv = cc->b[b], best = b;
                char ch = draft;
                if (best >= 0) { if (dcode < 4 && cc->b[dcode] == bv) best = dcode; ch = "ACGT"[best]; }
                if (w < cap) out[w] = ch;
                ++w;
            }
            for (int k = 0; k < CONS_KMAX && 2 * cc->insn[k] > cov; ++k) {
                int best = -1; uint32_t bv = 0;
                for (int b = 0; b < 4; ++b) if (cc->insb[k][b] > bv) bv = cc->insb[k][b], best = b;
                if (w < cap) out[w] = best >= 0 ? "ACGT"[best] : 'N';
                ++w;
            }
        }
        out_len[t] = (int32_t)(w - out_off[t]);
    }
    free(cell); free(base);
    return w;
}

/* ------------------------------------------------------------------------- */
/* 9. window partial-order consensus (spec 3.13; SURVEY 8(f) rank 4: the polishing hand-off H3, TELR_assembly.py:226-247 pipes
 *    `samtools view -F0x900` of the reads->contig alignments into wtpoa-cns).  wtpoa-cns's source is absent, so this restates the
 *    PUBLISHED scheme of window POA polishing (Lee 2002 Bioinformatics 18:452 partial-order alignment; Vaser 2017 Genome Res
 *    27:737 window consensus): the draft is cut into windows of POA_W bases; every primary record that covers a window whole gives
 *    the piece of its read that its CIGAR aligns to the window; the pieces are aligned one after the other to a graph that starts
 *    as the draft's window (global sequence-to-graph alignment, linear gap) and are merged into it; the consensus of the window
 *    is the heaviest-bundle path.  Every cap and tie-break below is part of the spec (the HIP kernel follows it bit for bit). */
#define POA_W       200     /* window (draft bases) */
#define POA_SEGMAX  400     /* a piece longer than 2 W or shorter than W / 2 is a structural difference, not an error of the draft */
#define POA_MAXSEG  64      /* pieces per window (the first ones in record order) */
#define POA_MAXNODE 2048
#define POA_MAXIN   8       /* in-edges per node; an edge beyond that is not recorded */
#define POA_BAND    64      /* cells per node row: the piece's positions within ~32 of where the node's column falls on the piece */
#define POA_MAXINDEL 30     /* a longer D / I run is a structural difference: its piece does not vote (CONS_MAXDEL of the pile-up) */
#define POA_M       3
#define POA_X       (-5)
#define POA_G       (-4)
typedef struct {
    int n;                                   /* nodes */
    uint8_t base[POA_MAXNODE];
    int16_t nin[POA_MAXNODE], nout[POA_MAXNODE];
    int16_t in[POA_MAXNODE][POA_MAXIN]; int16_t inw[POA_MAXNODE][POA_MAXIN];
    int16_t ring[POA_MAXNODE];               /* next node aligned to the same column (circular; itself when alone) */
    int16_t order[POA_MAXNODE];              /* topological order */
    int16_t startc[POA_MAXNODE], endc[POA_MAXNODE];      /* sequences (the draft's window included) that begin / end at the node */
    int16_t col[POA_MAXNODE];                /* the window column the node belongs to (a new node: that of the node it is aligned to / put behind) */
    int L;                                   /* columns of the window */
} poa_t;
static void poa_add_edge(poa_t *g, int u, int v)
{
    for (int k = 0; k < g->nin[v]; ++k) if (g->in[v][k] == u) { ++g->inw[v][k]; return; }
    if (g->nin[v] >= POA_MAXIN) return;
    g->in[v][g->nin[v]] = (int16_t)u; g->inw[v][g->nin[v]] = 1; ++g->nin[v]; ++g->nout[u];
}
/* align seq[0..n) to the graph and merge it in; H is scratch of (nodes + 1) x (n + 1) */
/* The score matrix is BANDED: the row of a node holds POA_BAND cells, the piece's positions [lo, lo + POA_BAND) with
 * lo = clamp((col + 1) * n / L - POA_BAND / 2, 0, max(0, n + 1 - POA_BAND)) -- around where the node's column falls on the piece
 * when piece and window are stretched onto each other (a 200-base window of a 10 %-error read drifts by a few bases, never 32);
 * every cell outside a row's band counts as -32000.  Row 0 (the virtual start) is j * gap and not stored. */
static inline int poa_lo(int col, int n, int L) { int lo = (col + 1) * n / L - POA_BAND / 2, hi = n + 1 - POA_BAND; if (lo > hi) lo = hi; return lo < 0 ? 0 : lo; }
static inline int poa_cell(const int16_t *H, const int16_t *lo, int row, int j, int n)
{
    if (row == 0) return j * POA_G;
    const int jj = j - lo[row - 1];
    return (jj < 0 || jj >= POA_BAND || j > n) ? -32000 : H[(size_t)(row - 1) * POA_BAND + jj];
}
static void poa_add_seq(poa_t *g, const uint8_t *seq, int n, int16_t *H, int16_t *rank)
{
    static __thread int16_t lo[POA_MAXNODE];
    for (int r = 0; r < g->n; ++r) { rank[g->order[r]] = (int16_t)(r + 1); lo[r] = (int16_t)poa_lo(g->col[g->order[r]], n, g->L); }
    for (int r = 0; r < g->n; ++r) {
        const int v = g->order[r];
        int16_t *row = H + (size_t)r * POA_BAND;
        const int j0 = lo[r], j1 = j0 + POA_BAND - 1 < n ? j0 + POA_BAND - 1 : n;
        for (int j = j0; j <= j1; ++j) {
            int best = -32000;
            const int npred = g->nin[v] ? g->nin[v] : 1;
            for (int k = 0; k < npred; ++k) {
                const int pr = g->nin[v] ? rank[g->in[v][k]] : 0;
                int c = poa_cell(H, lo, pr, j, n) + POA_G;                                  /* the node is skipped */
                if (c > best) best = c;
                if (j > 0) { c = poa_cell(H, lo, pr, j - 1, n) + (seq[j - 1] == g->base[v] && seq[j - 1] < 4 ? POA_M : POA_X); if (c > best) best = c; }
            }
            if (j > j0) { const int c = row[j - 1 - j0] + POA_G; if (c > best) best = c; }  /* the base is inserted */
            row[j - j0] = (int16_t)best;
        }
    }
    /* the end: the sink with the best score, smallest id on ties */
    int endv = -1, endsc = -32768;
    for (int v = 0; v < g->n; ++v) if (!g->nout[v]) { const int sc = poa_cell(H, lo, rank[v], n, n); if (sc > endsc) endsc = sc, endv = v; }
    /* walk back: diagonal from the first pred that explains the cell, else skip-node from the first pred, else inserted base */
    static __thread int16_t pn[POA_MAXNODE + POA_SEGMAX + 2], pj[POA_MAXNODE + POA_SEGMAX + 2];
    int np = 0, v = endv, j = n;
    while (v >= 0 || j > 0) {
        if (v < 0) { pn[np] = -1; pj[np] = (int16_t)(j - 1); ++np; --j; continue; }          /* bases before the graph's start */
        const int cur = poa_cell(H, lo, rank[v], j, n), npred = g->nin[v] ? g->nin[v] : 1;
        int moved = 0;
        if (j > 0) {
            const int sc = seq[j - 1] == g->base[v] && seq[j - 1] < 4 ? POA_M : POA_X;
            for (int k = 0; k < npred && !moved; ++k) {
                const int p = g->nin[v] ? g->in[v][k] : -1;
                if (poa_cell(H, lo, p >= 0 ? rank[p] : 0, j - 1, n) + sc == cur) { pn[np] = (int16_t)v; pj[np] = (int16_t)(j - 1); ++np; v = p; --j; moved = 1; }
            }
        }
        for (int k = 0; k < npred && !moved; ++k) {
            const int p = g->nin[v] ? g->in[v][k] : -1;
            if (poa_cell(H, lo, p >= 0 ? rank[p] : 0, j, n) + POA_G == cur) { v = p; moved = 1; }      /* node skipped: nothing to merge */
        }
        if (!moved) { pn[np] = -1; pj[np] = (int16_t)(j - 1); ++np; --j; }
    }
    /* merge, start -> end, keeping `order` topological WITHOUT re-sorting.  Invariant: the nodes of one column (a ring) are
     * contiguous in `order`, the column's first node in front.  A new node aligned to column x (another base there) goes right
     * behind x; a new node that stands for an inserted base goes behind the LAST node of the column before it (or keeps the place
     * of the new node before it; a new first node goes in front of everything).  anchor[k] = the old rank the k-th new node is
     * put behind (-1: the front); anchors do not decrease along the path, nodes with one anchor keep their path order. */
    static __thread int16_t newv[POA_SEGMAX + 2], anchor[POA_SEGMAX + 2];
    const int n_old = g->n;
    int prev = -1, nnew = 0, behind = -1;                                  /* behind: where an inserted base would go now */
    for (int z = np - 1; z >= 0; --z) {
        const uint8_t b = seq[pj[z]];
        int u = -1;
        const int x = pn[z];
        if (x >= 0) {
            if (g->base[x] == b) u = x;
            else for (int s_ = g->ring[x]; s_ != x; s_ = g->ring[s_]) if (g->base[s_] == b) { u = s_; break; }
        }
        if (u < 0) {
            u = g->n++;
            g->base[u] = b; g->nin[u] = g->nout[u] = 0; g->ring[u] = (int16_t)u; g->startc[u] = g->endc[u] = 0;
            g->col[u] = x >= 0 ? g->col[x] : prev >= 0 ? g->col[prev] : 0;
            newv[nnew] = (int16_t)u; anchor[nnew] = (int16_t)(x >= 0 ? rank[x] - 1 : behind); ++nnew;      /* rank[] is 1-based */
            if (x >= 0) { g->ring[u] = g->ring[x]; g->ring[x] = (int16_t)u; }
        }
        if (x >= 0) {                                                      /* behind the whole column of x (its old members) */
            int m = rank[x] - 1;
            for (int s_ = g->ring[x]; s_ != x; s_ = g->ring[s_]) if (s_ < n_old && rank[s_] - 1 > m) m = rank[s_] - 1;
            behind = m;
        }
        if (prev >= 0) poa_add_edge(g, prev, u); else ++g->startc[u];
        prev = u;
    }
    if (prev >= 0) ++g->endc[prev];
    if (nnew) {
        static __thread int16_t no[POA_MAXNODE];
        const int nold = g->n - nnew;
        int k = 0, o = 0;
        while (k < nnew && anchor[k] < 0) no[o++] = newv[k++];
        for (int i = 0; i < nold; ++i) { no[o++] = g->order[i]; while (k < nnew && anchor[k] == i) no[o++] = newv[k++]; }
        memcpy(g->order, no, sizeof(int16_t) * (size_t)g->n);
    }
    /* the checker checks itself: every edge must point forward in `order` (the sweep of the next piece relies on it) */
    for (int r = 0; r < g->n; ++r) rank[g->order[r]] = (int16_t)(r + 1);
    for (int v = 0; v < g->n; ++v) for (int k = 0; k < g->nin[v]; ++k) if (rank[g->in[v][k]] >= rank[v]) { fprintf(stderr, "tor_poa: edge %d -> %d against the order\n", g->in[v][k], v); abort(); }
}
/* heaviest bundle: every node takes its heaviest in-edge (the better-scored source, then the first, on ties).  The consensus
 * ENDS at the node most sequences end at and BEGINS at the node most sequences begin at (better score / smaller id on ties; the
 * walk back stops there, or at a node without in-edges): a piece whose cut at the window border is one base off must not
 * lengthen the window's consensus by that base -> bases in out[], returns the length */
static int poa_consensus(const poa_t *g, char *out)
{
    static __thread int32_t score[POA_MAXNODE]; static __thread int16_t bp[POA_MAXNODE];
    for (int r = 0; r < g->n; ++r) {
        const int v = g->order[r];
        int bw = -1, bs = -1, b = -1;
        for (int k = 0; k < g->nin[v]; ++k) {
            const int u = g->in[v][k], w = g->inw[v][k];
            if (w > bw || (w == bw && score[u] > bs)) { bw = w; bs = score[u]; b = u; }
        }
        bp[v] = (int16_t)b; score[v] = b >= 0 ? bs + bw : 0;
    }
    int endv = -1, startv = -1;
    for (int v = 0; v < g->n; ++v) {
        if (endv < 0 || g->endc[v] > g->endc[endv] || (g->endc[v] == g->endc[endv] && score[v] > score[endv])) endv = v;
        if (startv < 0 || g->startc[v] > g->startc[startv]) startv = v;
    }
    static __thread char tmp[POA_MAXNODE];
    int n = 0;
    for (int v = endv; v >= 0; v = bp[v]) { tmp[n++] = "ACGTN"[g->base[v]]; if (v == startv) break; }
    for (int i = 0; i < n; ++i) out[i] = tmp[n - 1 - i];
    return n;
}
/* same interface as tor_consensus */
int64_t tor_poa(const telr_aln *alns, int64_t n_aln, const uint32_t *cigars, const uint8_t *q_nt4, const int64_t *qoff,
                int32_t n_targets, const char *t_ascii, const int64_t *toff, const int32_t *tlen, int32_t min_depth,
                char *out, int64_t cap, int64_t *out_off, int32_t *out_len)
{
    poa_t *g = (poa_t*)malloc(sizeof(poa_t));
    int16_t *H = (int16_t*)malloc(sizeof(int16_t) * (size_t)POA_MAXNODE * POA_BAND), *rank = (int16_t*)malloc(2 * POA_MAXNODE);
    uint8_t *seg = (uint8_t*)malloc((size_t)POA_MAXSEG * POA_SEGMAX); int seglen[POA_MAXSEG];
    char *wout = (char*)malloc(POA_MAXNODE);
    /* records by target, in record order */
    int64_t *first = (int64_t*)calloc(n_targets + 1, 8), *idx = (int64_t*)malloc(8 * (n_aln ? n_aln : 1));
    for (int64_t i = 0; i < n_aln; ++i) if (!(alns[i].flags & (TELR_F_SECONDARY | TELR_F_SUPPL))) ++first[alns[i].tid + 1];
    for (int32_t t = 0; t < n_targets; ++t) first[t + 1] += first[t];
    { int64_t *fill = (int64_t*)malloc(8 * (n_targets + 1)); memcpy(fill, first, 8 * (n_targets + 1));
      for (int64_t i = 0; i < n_aln; ++i) if (!(alns[i].flags & (TELR_F_SECONDARY | TELR_F_SUPPL))) idx[fill[alns[i].tid]++] = i;
      free(fill); }
    int64_t w = 0;
    for (int32_t t = 0; t < n_targets; ++t) {
        out_off[t] = w;
        const char *ts = t_ascii + toff[t];
        for (int32_t w0 = 0; w0 < tlen[t]; w0 += POA_W) {
            const int32_t w1 = w0 + POA_W < tlen[t] ? w0 + POA_W : tlen[t];
            int nseg = 0;
            for (int64_t z = first[t]; z < first[t + 1] && nseg < POA_MAXSEG; ++z) {
                const telr_aln *r = &alns[idx[z]];
                if (r->ts > w0 || r->te < w1) continue;                     /* the record must cover the window whole */
                const int rev = (r->flags & TELR_F_REV) != 0;
                const uint8_t *q = q_nt4 + qoff[r->qid];
                /* the piece = query bases [qa, qb): qa / qb = the query offset where the target first reaches w0 / w1 (at the first op
                 * that consumes a target base beyond it), so that a read's pieces tile it and an insertion at a window border belongs to
                 * the window before it.  A D or I run longer than POA_MAXINDEL that touches the window is a structural difference (a read
                 * of the other allele), not an error of the draft: the piece does not vote (the pile-up's rule, spec 3.12). */
                int32_t qi = rev ? r->qlen - r->qe : r->qs, ti = r->ts, qa = -1, qb = -1, big = 0;
                for (int32_t c = 0; c < r->n_cigar && qb < 0; ++c) {
                    const uint32_t cg = cigars[r->cigar_off + c]; const int op = cg & 0xf, l = cg >> 4;
                    if (op == 1) { if (l > POA_MAXINDEL && ti > w0 && ti <= w1) big = 1; qi += l; continue; }
                    if (qa < 0 && w0 < ti + l) qa = op == 0 ? qi + (w0 - ti) : qi;
                    if (w1 < ti + l) qb = op == 0 ? qi + (w1 - ti) : qi;
                    if (op == 2 && l > POA_MAXINDEL && ti < w1 && ti + l > w0) big = 1;
                    if (op == 0) qi += l;
                    ti += l;
                }
                if (qb < 0) qb = qi;                                         /* the record ends with the window */
                if (big) continue;
                const int len = qb - qa;
                if (qa < 0 || len < (w1 - w0) / 2 || len > POA_SEGMAX) continue;     /* (a piece shorter than half the window: the read lacks what the draft has here) */
                int ok = 1;
                for (int x = 0; x < len; ++x) {
                    int b = rev ? q[r->qlen - 1 - (qa + x)] : q[qa + x];
                    if (rev && b < 4) b = 3 - b;
                    if (b > 3) { ok = 0; break; }                             /* a piece with an ambiguous base does not vote */
                    seg[(size_t)nseg * POA_SEGMAX + x] = (uint8_t)b;
                }
                if (!ok) continue;
                seglen[nseg++] = len;
            }
            if (nseg < min_depth) { for (int32_t p = w0; p < w1; ++p) { if (w < cap) out[w] = "ACGTN"[NT4[(uint8_t)ts[p]]]; ++w; } continue; }
            /* the graph starts as the draft's window */
            g->n = g->L = w1 - w0;
            for (int v = 0; v < g->n; ++v) {
                g->base[v] = NT4[(uint8_t)ts[w0 + v]]; g->nin[v] = g->nout[v] = 0; g->ring[v] = (int16_t)v; g->order[v] = (int16_t)v;
                g->startc[v] = v == 0; g->endc[v] = v == g->n - 1; g->col[v] = (int16_t)v;
                if (v) { g->in[v][0] = (int16_t)(v - 1); g->inw[v][0] = 1; g->nin[v] = 1; g->nout[v - 1] = 1; }
            }
            for (int s = 0; s < nseg; ++s) {
                if (g->n + seglen[s] > POA_MAXNODE) continue;                 /* the graph could outgrow its arrays: the piece is left out */
                poa_add_seq(g, seg + (size_t)s * POA_SEGMAX, seglen[s], H, rank);
            }
            const int cl = poa_consensus(g, wout);
            for (int x = 0; x < cl; ++x) { if (w < cap) out[w] = wout[x]; ++w; }
        }
        out_len[t] = (int32_t)(w - out_off[t]);
    }
    free(g); free(H); free(rank); free(seg); free(wout); free(first); free(idx);
    return w;
}
