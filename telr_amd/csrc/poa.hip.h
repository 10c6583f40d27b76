// telr_amd/csrc/poa.hip.h -- window partial-order consensus on the device (spec 3.13; oracle/telr_oracle.c: tor_poa).
//
// the polishing hand-off H3 (TELR_assembly.py:226-247 pipes `samtools view -F0x900` of the reads->contig
// alignments into wtpoa-cns).  wtpoa-cns is absent, so this is the published scheme of window POA polishing (Lee 2002; Vaser
// 2017), not its code: the draft is cut into windows of POA_W bases; every primary record that covers a window whole gives the
// piece of its read that its CIGAR aligns to the window; the pieces are aligned one after the other to a graph that starts as
// the draft's window (global sequence-to-graph alignment, linear gap) and merged into it; the window's consensus is the
// heaviest-bundle path between the nodes most sequences begin / end at.  Opt-in (`polish="poa"`): another consensus can change
// call sets.  Caps and tie-breaks are the oracle's, bit for bit (tests/test_gpu_consensus.py).
//
//   host            the pieces of every window from the records' CIGARs -- ONE walk per record gives the query offsets at all the
//                   window borders it crosses -- in record order, CSR by window
//   k_poa_window    ONE WAVE per window (persistent over the window list): the graph and the score matrix live in a slot of
//                   global scratch.  The matrix is BANDED: a node's row holds the POA_BAND = 64 columns around the diagonal
//                   of the node's window column (d_poa_lo; a piece is at most 30 bases off the diagonal by the POA_MAXINDEL
//                   rule), i.e. exactly one cell per lane.  The row of the node before is read from registers (lane shuffle),
//                   the in-row gap chain is a max-plus prefix scan over the wave, every cell notes where its value came from;
//                   the walk back (one byte per step) and the heaviest-bundle pass are serial stretches on lane 0, the merge of
//                   a piece into the graph and the new topological order are parallel passes over the path.
//   k_poa_pack      the windows' strings, packed, for the copy back
// Integer work bounded by instruction issue and L2 latency: no MFMA.
#pragma once

#define POA_W       200
#define POA_SEGMAX  400
#define POA_MAXSEG  64
#define POA_MAXNODE 2048
#define POA_MAXIN   8
#define POA_MAXINDEL 30
#define POA_BAND    64      /* cells per node row (oracle: poa_lo / poa_cell) */
#define POA_M       3
#define POA_X       (-5)
#define POA_G       (-4)

struct PoaPiece { int32_t qid, qa, len, rev; };           // query bases [qa, qa + len) on the alignment strand
struct PoaArgs {
    const PoaPiece *pieces; const int32_t *wptr;          // pieces of window w: [wptr[w], wptr[w + 1]) in record order
    const int32_t *w_tid, *w_w0, *w_w1; int32_t nwin, min_depth;
    const uint32_t *q2, *qn; const int64_t *qboff; const int32_t *qlen;
    const uint32_t *t2, *tn; const int64_t *tboff;
    uint8_t *scratch; size_t slot_bytes;                  // one slot per resident wave
    uint8_t *wout; int32_t *wlen;                         // consensus of window w: wout[w * POA_MAXNODE ...], wlen[w]
};
// slot layout (bytes)
#define POA_O_BASE   0                                    /* u8  [MAXNODE] */
#define POA_O_NIN    (POA_O_BASE + POA_MAXNODE)           /* u8  [MAXNODE] */
#define POA_O_NOUT   (POA_O_NIN + POA_MAXNODE)            /* i16 [MAXNODE] */
#define POA_O_IN     (POA_O_NOUT + 2 * POA_MAXNODE)       /* i16 [MAXNODE][MAXIN] */
#define POA_O_INW    (POA_O_IN + 2 * POA_MAXNODE * POA_MAXIN)
#define POA_O_RING   (POA_O_INW + 2 * POA_MAXNODE * POA_MAXIN)
#define POA_O_ORDER  (POA_O_RING + 2 * POA_MAXNODE)
#define POA_O_RANK   (POA_O_ORDER + 2 * POA_MAXNODE)
#define POA_O_STARTC (POA_O_RANK + 2 * POA_MAXNODE)
#define POA_O_ENDC   (POA_O_STARTC + 2 * POA_MAXNODE)
#define POA_O_BP     (POA_O_ENDC + 2 * POA_MAXNODE)
#define POA_O_NO     (POA_O_BP + 2 * POA_MAXNODE)
#define POA_O_SCORE  (POA_O_NO + 2 * POA_MAXNODE)         /* i32 [MAXNODE] */
#define POA_O_PN     (POA_O_SCORE + 4 * POA_MAXNODE)      /* i16 [MAXNODE + SEGMAX + 2] */
#define POA_NPATH    (POA_MAXNODE + POA_SEGMAX + 8)
#define POA_O_PJ     (POA_O_PN + 2 * POA_NPATH)
#define POA_O_NEWV   (POA_O_PJ + 2 * POA_NPATH)           /* i16 [SEGMAX + 8] */
#define POA_O_ANCH   (POA_O_NEWV + 2 * (POA_SEGMAX + 8))
#define POA_O_PROW   ((POA_O_ANCH + 2 * (POA_SEGMAX + 8) + 15) & ~15)       /* i16 [MAXNODE][MAXIN]: rows of the predecessors of the node at rank r */
#define POA_O_PPLO   (POA_O_PROW + 2 * POA_MAXNODE * POA_MAXIN)             /* i16 [MAXNODE][MAXIN]: first column of those rows' bands */
#define POA_O_PCB    (POA_O_PPLO + 2 * POA_MAXNODE * POA_MAXIN)             /* i16 [MAXNODE]: predecessors << 8 | base of the node at rank r */
#define POA_O_COL    (POA_O_PCB + 2 * POA_MAXNODE)                          /* i16 [MAXNODE]: window column of the node */
#define POA_O_LO     (POA_O_COL + 2 * POA_MAXNODE)                          /* i16 [MAXNODE]: first column of the band of the node at rank r */
#define POA_O_H      ((POA_O_LO + 2 * POA_MAXNODE + 255) & ~255)            /* i16 [MAXNODE][POA_BAND]: the banded score matrix (row r = the node at rank r) */
#define POA_O_TB     (POA_O_H + 2 * POA_MAXNODE * POA_BAND)                 /* u8  [MAXNODE][POA_BAND]: where a cell's value came from (0-7: diagonal from predecessor k, 8-15: skipped the node, from predecessor k, 16: base inserted) */
#define POA_SLOT_BYTES ((size_t)POA_O_TB + (size_t)POA_MAXNODE * POA_BAND)

__host__ __device__ __forceinline__ int d_poa_lo(int col, int n, int L) { int lo = (col + 1) * n / L - POA_BAND / 2, hi = n + 1 - POA_BAND; if (lo > hi) lo = hi; return lo < 0 ? 0 : lo; }
// cell (row, j) of the banded matrix: row 0 = the virtual start, outside a row's band -32000
__device__ __forceinline__ int d_poa_cell(const int16_t *H, const int16_t *lo, int row, int j, int n)
{
    if (row == 0) return j * POA_G;
    const int jj = j - lo[row - 1];
    return (jj < 0 || jj >= POA_BAND || j > n) ? -32000 : H[(size_t)(row - 1) * POA_BAND + jj];
}

__device__ __forceinline__ int d_poa_qbase(const PoaArgs &A, const PoaPiece &P, int x)
{
    // base x of the piece on the alignment strand (a reverse record reads its query mirrored and complemented)
    const int64_t b0 = A.qboff[P.qid]; const int ql = A.qlen[P.qid];
    const int pos = P.rev ? ql - 1 - (P.qa + x) : P.qa + x;
    int b = d_base(A.q2, A.qn, b0 + pos);
    if (P.rev && b < 4) b = 3 - b;
    return b;
}

__global__ void __launch_bounds__(64) k_poa_window(PoaArgs A)
{
    __shared__ uint8_t seq[POA_SEGMAX + 8];
    __shared__ int32_t sel[POA_MAXSEG];
    __shared__ int32_t sh[8];
    const int lane = threadIdx.x;
    uint8_t *S = A.scratch + (size_t)blockIdx.x * A.slot_bytes;
    uint8_t *base = S + POA_O_BASE, *nin = S + POA_O_NIN;
    int16_t *nout = (int16_t*)(S + POA_O_NOUT), *in = (int16_t*)(S + POA_O_IN), *inw = (int16_t*)(S + POA_O_INW), *ring = (int16_t*)(S + POA_O_RING);
    int16_t *order = (int16_t*)(S + POA_O_ORDER), *rank = (int16_t*)(S + POA_O_RANK), *startc = (int16_t*)(S + POA_O_STARTC), *endc = (int16_t*)(S + POA_O_ENDC);
    int16_t *bp = (int16_t*)(S + POA_O_BP), *no = (int16_t*)(S + POA_O_NO), *pn = (int16_t*)(S + POA_O_PN), *pj = (int16_t*)(S + POA_O_PJ);
    int16_t *newv = (int16_t*)(S + POA_O_NEWV), *anchor = (int16_t*)(S + POA_O_ANCH), *H = (int16_t*)(S + POA_O_H);
    int32_t *score = (int32_t*)(S + POA_O_SCORE);
    uint8_t *TB = S + POA_O_TB;
    int16_t *prow = (int16_t*)(S + POA_O_PROW), *pplo = (int16_t*)(S + POA_O_PPLO), *pcb = (int16_t*)(S + POA_O_PCB), *col = (int16_t*)(S + POA_O_COL), *lo_r = (int16_t*)(S + POA_O_LO);
    for (int w = blockIdx.x; w < A.nwin; w += gridDim.x) {
        const int tid = A.w_tid[w], w0 = A.w_w0[w], w1 = A.w_w1[w], L = w1 - w0;
        const int64_t tb0 = A.tboff[tid];
        uint8_t *out = A.wout + (size_t)w * POA_MAXNODE;
        // ---- the pieces that vote: the first POA_MAXSEG without an ambiguous base, in record order
        int nsel = 0;
        for (int c = A.wptr[w]; c < A.wptr[w + 1] && nsel < POA_MAXSEG; ++c) {
            const PoaPiece P = A.pieces[c];
            bool bad = false;
            for (int x = lane; x < P.len; x += 64) bad |= d_poa_qbase(A, P, x) > 3;
            if (!__any(bad)) { if (lane == 0) sel[nsel] = c; ++nsel; }
        }
        __syncthreads();
        if (nsel < A.min_depth) {
            for (int x = lane; x < L; x += 64) out[x] = "ACGTN"[d_base(A.t2, A.tn, tb0 + w0 + x)];
            if (lane == 0) A.wlen[w] = L;
            __syncthreads();
            continue;
        }
        // ---- the graph starts as the draft's window
        int n = L;
        for (int v = lane; v < L; v += 64) {
            base[v] = (uint8_t)d_base(A.t2, A.tn, tb0 + w0 + v); nin[v] = v ? 1 : 0; nout[v] = v < L - 1 ? 1 : 0; ring[v] = (int16_t)v; order[v] = (int16_t)v;
            startc[v] = v == 0; endc[v] = v == L - 1; col[v] = (int16_t)v;
            if (v) { in[v * POA_MAXIN] = (int16_t)(v - 1); inw[v * POA_MAXIN] = 1; }
        }
        __syncthreads();
        for (int si = 0; si < nsel; ++si) {
            const PoaPiece P = A.pieces[sel[si]];
            const int m = P.len;
            if (n + m > POA_MAXNODE) continue;
            for (int x = lane; x < m; x += 64) seq[x] = (uint8_t)d_poa_qbase(A, P, x);
            for (int r = lane; r < n; r += 64) { const int v = order[r]; rank[v] = (int16_t)(r + 1); lo_r[r] = (int16_t)d_poa_lo(col[v], m, L); }
            __syncthreads();
            // per RANK: the rows of the node's predecessors, the first columns of their bands and (count << 8 | base) -- the sweep reads
            // sequential tables, one node ahead, instead of chasing order -> in-list -> rank inside every step
            for (int r = lane; r < n; r += 64) {
                const int v = order[r], c = nin[v];
                pcb[r] = (int16_t)((c ? c : 1) << 8 | base[v]);
                for (int k = 0; k < (c ? c : 1); ++k) {
                    const int pr = c ? rank[in[v * POA_MAXIN + k]] : 0;
                    prow[r * POA_MAXIN + k] = (int16_t)pr; pplo[r * POA_MAXIN + k] = pr ? lo_r[pr - 1] : 0;
                }
            }
            __syncthreads();
            // ---- sweep: one row of POA_BAND cells per node, in topological order: lane l computes column lo + l.  The row of the
            // node before (nearly every node's only predecessor) stays in registers and is read with a lane shuffle -- waiting for the
            // row just stored to come back from L2 was most of a step; rows further back are read from the slot.  Every cell also
            // notes where its value came from, by the walk's own preference (diagonal before skipped node before inserted base, the
            // first predecessor that explains it): the walk back then reads one byte per step instead of all candidates again.
            int nx_cb = pcb[0], nx_lo = lo_r[0]; uint32_t nx_p01 = *(const uint32_t*)&prow[0], nx_l01 = *(const uint32_t*)&pplo[0];
            int prev_val = -32000, prev_lo = 0;
            for (int r = 0; r < n; ++r) {
                const int cb = __builtin_amdgcn_readfirstlane(nx_cb), np_ = cb >> 8, vb = cb & 0xff, lo = __builtin_amdgcn_readfirstlane(nx_lo), j = lo + lane;
                const uint32_t p01 = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx_p01), l01 = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx_l01);
                if (r + 1 < n) { nx_cb = pcb[r + 1]; nx_lo = lo_r[r + 1]; nx_p01 = *(const uint32_t*)&prow[(r + 1) * POA_MAXIN]; nx_l01 = *(const uint32_t*)&pplo[(r + 1) * POA_MAXIN]; }
                const int sb = j > 0 && j <= m ? seq[j - 1] : 4;
                const int sc = sb == vb && sb < 4 ? POA_M : POA_X;
                int vmax = -1000000, vk = 0, dmax = -1000000, dk = 0;
                bool synced = false;
                for (int k = 0; k < np_; ++k) {
                    const int pr = k == 0 ? (int)(p01 & 0xffffu) : k == 1 ? (int)(p01 >> 16) : __builtin_amdgcn_readfirstlane((int)prow[r * POA_MAXIN + k]);
                    const int pl = k == 0 ? (int)(l01 & 0xffffu) : k == 1 ? (int)(l01 >> 16) : __builtin_amdgcn_readfirstlane((int)pplo[r * POA_MAXIN + k]);
                    int vj, vj1;
                    if (pr == 0) { vj = j * POA_G; vj1 = (j - 1) * POA_G; }
                    else if (pr == r) {                          // the row before: registers
                        const int jj = j - prev_lo;
                        const int a = __shfl(prev_val, jj & 63), b = __shfl(prev_val, (jj - 1) & 63);
                        vj = (jj >= 0 && jj < POA_BAND) ? a : -32000;
                        vj1 = (jj >= 1 && jj <= POA_BAND) ? b : -32000;
                    } else {
                        if (!synced) { __syncthreads(); synced = true; }        // (rows stored by other lanes of this wave)
                        const int jj = j - pl;
                        const int16_t *prw = H + (size_t)(pr - 1) * POA_BAND;
                        vj = (jj >= 0 && jj < POA_BAND && j <= m) ? prw[jj] : -32000;
                        vj1 = (jj >= 1 && jj <= POA_BAND && j <= m) ? prw[jj - 1] : -32000;
                    }
                    int c = vj + POA_G; if (c > vmax) { vmax = c; vk = k; }
                    c = vj1 + sc; if (j > 0 && c > dmax) { dmax = c; dk = k; }
                }
                int t = vmax > dmax ? vmax : dmax; t = t > -32000 ? t : -32000;
                // row[j] = max over the band's k <= j of T[k] + (j - k) G  =  (prefix max of T[k] - k G) + j G
                int u = j <= m ? t - j * POA_G : -1000000;
#pragma unroll
                for (int s_ = 1; s_ < 64; s_ <<= 1) { const int o = __shfl_up(u, s_); if (lane >= s_) u = o > u ? o : u; }
                const int cur = (int)(int16_t)(u + j * POA_G);
                if (j <= m) {
                    H[(size_t)r * POA_BAND + lane] = (int16_t)cur;
                    TB[(size_t)r * POA_BAND + lane] = (uint8_t)(j > 0 && dmax == cur ? dk : vmax == cur ? 8 | vk : 16);
                }
                prev_val = j <= m ? cur : -32000; prev_lo = lo;
            }
            __syncthreads();
            // ---- the end: the node without out-edges whose last column scores best, smallest id on ties
            {
                int bs = -32768, bv = 0x7fffffff;
                for (int v = lane; v < n; v += 64) if (!nout[v]) { const int sc = d_poa_cell(H, lo_r, rank[v], m, m); if (sc > bs) { bs = sc; bv = v; } }
#pragma unroll
                for (int s = 32; s >= 1; s >>= 1) { const int os = __shfl_xor(bs, s), ov = __shfl_xor(bv, s); if (os > bs || (os == bs && ov < bv)) { bs = os; bv = ov; } }
                if (lane == 0) sh[0] = bv == 0x7fffffff ? -1 : bv;
            }
            __syncthreads();
            if (lane == 0) {
                // ---- walk back along the notes of the sweep (pn holds ROW numbers here, 0 = no node; turned into nodes below)
                int np = 0, j = m;
                int r = sh[0] >= 0 ? rank[sh[0]] : 0, lo = r ? lo_r[r - 1] : 0;
                for (int it = 0; (r > 0 || j > 0) && j >= 0 && it < 2 * POA_NPATH; ++it) {        // (the bounds only keep a broken slot from hanging the device)
                    const int jj = j - lo;
                    const int tb = r > 0 && jj >= 0 && jj < POA_BAND ? TB[(size_t)(r - 1) * POA_BAND + jj] : 16;
                    if (tb < 16) {
                        const int k = tb & 7, nr = prow[(r - 1) * POA_MAXIN + k], nlo = pplo[(r - 1) * POA_MAXIN + k];
                        if (tb < 8) { pn[np] = (int16_t)r; pj[np] = (int16_t)(j - 1); ++np; --j; }
                        r = nr; lo = nlo;
                    } else { pn[np] = 0; pj[np] = (int16_t)(j - 1); ++np; --j; }
                }
                sh[3] = np;
            }
            __syncthreads();
            for (int z = lane; z < sh[3]; z += 64) { const int r = pn[z]; pn[z] = r ? order[r - 1] : (int16_t)-1; }
            __syncthreads();
            // ---- merge, start -> end (placement rules: see the oracle), as a PARALLEL pass over the path: step t (lane t of a chunk of
            // 64) touches only its own node -- a path visits a column once, so the rings searched / extended by two steps are
            // disjoint, every step's target node u_t and source node u_(t-1) are distinct from every other step's -- and what the serial
            // loop carries from step to step are three "last value so far" scans: the id of the next new node (a count), and for
            // inserted bases the rank bound `behind` and the column of the last aligned step before them.
            {
                const int np = sh[3], n_old = n;
                int c_new = 0, c_behind = -1, c_col = 0, c_prevu = -1;
                const uint64_t below = (1ULL << lane) - 1ULL;
                for (int t0 = 0; t0 < np; t0 += 64) {
                    const int t = t0 + lane, z = np - 1 - t; const bool on = t < np;
                    int x = -1, u = -1, mm = -1, cx = 0; uint8_t b = 4;
                    if (on) { x = pn[z]; b = seq[pj[z]]; }
                    if (x >= 0) {
                        if (base[x] == b) u = x;
                        else for (int s_ = ring[x]; s_ != x; s_ = ring[s_]) if (base[s_] == b) { u = s_; break; }
                        mm = rank[x] - 1;
                        for (int s_ = ring[x]; s_ != x; s_ = ring[s_]) if (s_ < n_old && rank[s_] - 1 > mm) mm = rank[s_] - 1;
                        cx = col[x];
                    }
                    const bool isnew = on && u < 0;
                    const uint64_t al = __ballot(x >= 0), nw = __ballot(isnew), alb = al & below;
                    const int src = alb ? 63 - __builtin_clzll(alb) : 0;
                    const int mm_b = __shfl(mm, src), cx_b = __shfl(cx, src);
                    const int behind_before = alb ? mm_b : c_behind, col_ins = alb ? cx_b : c_col;
                    const int k = c_new + (int)__popcll(nw & below);
                    if (isnew) {
                        u = n_old + k;
                        base[u] = b; nin[u] = 0; nout[u] = 0; startc[u] = 0; endc[u] = 0;
                        col[u] = (int16_t)(x >= 0 ? cx : col_ins);
                        newv[k] = (int16_t)u; anchor[k] = (int16_t)(x >= 0 ? rank[x] - 1 : behind_before);
                        if (x >= 0) { ring[u] = ring[x]; ring[x] = (int16_t)u; } else ring[u] = (int16_t)u;
                    }
                    int pu = __shfl_up(u, 1); if (lane == 0) pu = c_prevu;
                    __syncthreads();                                    // the new nodes' fields, before their neighbours' edges
                    if (on) {
                        if (pu >= 0) {
                            bool found = false;
                            for (int q_ = 0; q_ < nin[u]; ++q_) if (in[u * POA_MAXIN + q_] == pu) { ++inw[u * POA_MAXIN + q_]; found = true; break; }
                            if (!found && nin[u] < POA_MAXIN) { in[u * POA_MAXIN + nin[u]] = (int16_t)pu; inw[u * POA_MAXIN + nin[u]] = 1; ++nin[u]; ++nout[pu]; }
                        } else ++startc[u];
                        if (t == np - 1) ++endc[u];
                    }
                    c_new += (int)__popcll(nw);
                    if (al) { const int last = 63 - __builtin_clzll(al); c_behind = __shfl(mm, last); c_col = __shfl(cx, last); }
                    { const int lastl = np - 1 - t0 < 63 ? np - 1 - t0 : 63; c_prevu = __shfl(u, lastl); }
                    __syncthreads();
                }
                const int nnew = c_new;
                n = n_old + nnew;
                if (nnew) {
                    // the order: new node k goes behind old rank anchor[k] (anchors do not decrease along the path), after the k new nodes before it
                    for (int k = lane; k < nnew; k += 64) no[anchor[k] + 1 + k] = newv[k];
                    for (int i_ = lane; i_ < n_old; i_ += 64) {
                        int lo_ = 0, hi_ = nnew;                        // anchors < i_
                        while (lo_ < hi_) { const int mid = (lo_ + hi_) >> 1; if (anchor[mid] < i_) lo_ = mid + 1; else hi_ = mid; }
                        no[i_ + lo_] = order[i_];
                    }
                    __syncthreads();
                    for (int i_ = lane; i_ < n; i_ += 64) order[i_] = no[i_];
                }
            }
            __syncthreads();
        }
        // ---- heaviest bundle between the most common start / end nodes
        if (lane == 0) {
            for (int r = 0; r < n; ++r) {
                const int v = order[r];
                int bw = -1, bs = -1, b = -1;
                for (int k = 0; k < nin[v]; ++k) {
                    const int u = in[v * POA_MAXIN + k], wgt = inw[v * POA_MAXIN + k];
                    if (wgt > bw || (wgt == bw && score[u] > bs)) { bw = wgt; bs = score[u]; b = u; }
                }
                bp[v] = (int16_t)b; score[v] = b >= 0 ? bs + bw : 0;
            }
            int endv = -1, startv = -1;
            for (int v = 0; v < n; ++v) {
                if (endv < 0 || endc[v] > endc[endv] || (endc[v] == endc[endv] && score[v] > score[endv])) endv = v;
                if (startv < 0 || startc[v] > startc[startv]) startv = v;
            }
            int cl = 0;
            for (int v = endv; v >= 0; v = bp[v]) { no[cl++] = (int16_t)v; if (v == startv) break; }
            for (int i = 0; i < cl; ++i) out[i] = "ACGTN"[base[no[cl - 1 - i]]];
            A.wlen[w] = cl;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(64) k_poa_pack(const uint8_t *__restrict__ wout, const int64_t *__restrict__ woff, uint8_t *__restrict__ pack)
{
    const int64_t w = blockIdx.x, o = woff[w]; const int n = (int)(woff[w + 1] - o);
    for (int i = threadIdx.x; i < n; i += 64) pack[o + i] = wout[(size_t)w * POA_MAXNODE + i];
}

// ---- host --------------------------------------------------------------------------------------------------------------
static int poa_impl(telr_ctx *ctx, const telr_result *r, const telr_seqset *queries, const telr_index *idx, int32_t min_depth, telr_consensus **out)
{
    if (!ctx || !r || !queries || !idx || !idx->targets || !out || min_depth < 0) return TELR_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    result_wait(r);
    hipStream_t st = ctx->stream;
    const telr_seqset *tg = idx->targets;
    const int32_t nt = tg->n;
    for (const telr_aln &a : r->alns) if (a.qid < 0 || a.qid >= queries->n || a.tid < 0 || a.tid >= nt) return TELR_E_ARG;
    // windows
    std::vector<int64_t> wbase((size_t)nt + 1, 0);
    for (int t = 0; t < nt; ++t) wbase[t + 1] = wbase[t] + (tg->len[t] + POA_W - 1) / POA_W;
    const int64_t nwin = wbase[nt];
    telr_consensus *C = new telr_consensus();
    C->off.assign((size_t)nt, 0); C->len.assign((size_t)nt, 0);
    if (nwin == 0) { *out = C; return TELR_OK; }
    if (nwin >= (1LL << 31) / POA_MAXNODE * 64) { delete C; return TELR_E_RANGE; }
    std::vector<int32_t> w_tid((size_t)nwin), w_w0((size_t)nwin), w_w1((size_t)nwin);
    for (int t = 0; t < nt; ++t) for (int64_t k = wbase[t]; k < wbase[t + 1]; ++k) {
        w_tid[k] = t; w_w0[k] = (int32_t)((k - wbase[t]) * POA_W); w_w1[k] = std::min(w_w0[k] + POA_W, tg->len[t]);
    }
    // the pieces, record by record (the walk of oracle/telr_oracle.c: tor_poa), then grouped by window keeping the record order
    struct Cand { int64_t win; PoaPiece p; };
    std::vector<Cand> cand;
    // ONE walk over a record's CIGAR serves all its windows (the oracle scans it once per window; same values): the query offset
    // at a window border P is that of the M / D op that contains P -- qi + (P - ti) inside an M, qi at a D -- or, when the record
    // ends exactly at P, the query offset behind its last op; an insertion longer than POA_MAXINDEL at ti spoils the window with
    // w0 < ti <= w1, such a deletion every window it overlaps.
    std::vector<int32_t> bval; std::vector<uint8_t> bbig;
    for (const telr_aln &a : r->alns) {
        if (a.flags & (TELR_F_SECONDARY | TELR_F_SUPPL)) continue;
        const bool rev = (a.flags & TELR_F_REV) != 0;
        const int32_t tl = tg->len[a.tid];
        const int32_t kf = (a.ts + POA_W - 1) / POA_W;                      // first window that starts inside the record
        int32_t cnt = 0;
        for (int32_t w0 = kf * POA_W; w0 < tl && a.te >= std::min(w0 + POA_W, tl); w0 += POA_W) ++cnt;
        if (cnt == 0) continue;
        // borders b = 0 .. cnt: positions (kf + b) W, the last one cut at the target's end
        bval.assign((size_t)cnt + 1, -1); bbig.assign((size_t)cnt, 0);
        auto border = [&](int32_t b) { return std::min((kf + b) * POA_W, tl); };
        int32_t qi = rev ? a.qlen - a.qe : a.qs, ti = a.ts, nb = 0;
        for (int32_t c = 0; c < a.n_cigar; ++c) {
            const uint32_t cg = r->cig[a.cigar_off + c]; const int op = cg & 0xf, l = (int)(cg >> 4);
            if (op == 1) {
                if (l > POA_MAXINDEL && ti > 0) { const int32_t k = (ti - 1) / POA_W - kf; if (k >= 0 && k < cnt && ti <= border(k + 1)) bbig[k] = 1; }
                qi += l; continue;
            }
            while (nb <= cnt && border(nb) < ti + l) { bval[nb] = op == 0 ? qi + (border(nb) - ti) : qi; ++nb; }
            if (op == 2 && l > POA_MAXINDEL) {
                for (int32_t k = std::max(0, ti / POA_W - kf); k < cnt && (kf + k) * POA_W < ti + l; ++k) if (ti < border(k + 1)) bbig[k] = 1;
            }
            if (op == 0) qi += l;
            ti += l;
        }
        for (int32_t k = 0; k < cnt; ++k) {
            const int32_t w0 = (kf + k) * POA_W, w1 = border(k + 1);
            const int32_t qa = bval[k], qb = bval[k + 1] >= 0 ? bval[k + 1] : qi;
            if (bbig[k]) continue;
            const int len = qb - qa;
            if (qa < 0 || len < (w1 - w0) / 2 || len > POA_SEGMAX) continue;
            Cand cd; cd.win = wbase[a.tid] + kf + k; cd.p.qid = a.qid; cd.p.qa = qa; cd.p.len = len; cd.p.rev = rev ? 1 : 0;
            cand.push_back(cd);
        }
    }
    std::vector<int32_t> wptr((size_t)nwin + 1, 0);
    for (const Cand &c : cand) ++wptr[c.win + 1];
    for (int64_t k = 0; k < nwin; ++k) wptr[k + 1] += wptr[k];
    std::vector<PoaPiece> pieces(cand.size());
    { std::vector<int32_t> fill(wptr.begin(), wptr.end() - 1); for (const Cand &c : cand) pieces[fill[c.win]++] = c.p; }
    // one slot per RESIDENT wave (a wave beyond that would start when the others have done all their windows: twice the time)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_poa_window, 64, 0) != hipSuccess || per_cu < 1) per_cu = 8;
    hipDeviceProp_t prop; int ncu = 256;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    const int64_t want = (int64_t)per_cu * ncu;
    const int nslot = (int)std::min<int64_t>(nwin, want);
    PoaArgs A; memset(&A, 0, sizeof(A));
    PoaPiece *d_p; int32_t *d_wptr, *d_wt, *d_w0, *d_w1, *d_wlen; uint8_t *d_scr, *d_wout;
    int rc;
    auto fail = [&](int code) { delete C; return code; };
    if ((rc = ctx_buf_t(ctx, "poa_pieces", pieces.size() + 1, &d_p)) != TELR_OK || (rc = ctx_buf_t(ctx, "poa_wptr", (size_t)nwin + 1, &d_wptr)) != TELR_OK ||
        (rc = ctx_buf_t(ctx, "poa_wt", (size_t)nwin, &d_wt)) != TELR_OK || (rc = ctx_buf_t(ctx, "poa_w0", (size_t)nwin, &d_w0)) != TELR_OK ||
        (rc = ctx_buf_t(ctx, "poa_w1", (size_t)nwin, &d_w1)) != TELR_OK || (rc = ctx_buf_t(ctx, "poa_wlen", (size_t)nwin, &d_wlen)) != TELR_OK ||
        (rc = ctx_buf_t(ctx, "poa_scratch", (size_t)nslot * POA_SLOT_BYTES, &d_scr)) != TELR_OK ||
        (rc = ctx_buf_t(ctx, "poa_wout", (size_t)nwin * POA_MAXNODE, &d_wout)) != TELR_OK) return fail(rc);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ctx->err = std::string(#x) + ": " + hipGetErrorString(e_); return fail(e_ == hipErrorOutOfMemory ? TELR_E_NOMEM : TELR_E_HIP); } } while (0)
    if (!pieces.empty()) CK(hipMemcpyAsync(d_p, pieces.data(), pieces.size() * sizeof(PoaPiece), hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(d_wptr, wptr.data(), ((size_t)nwin + 1) * 4, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(d_wt, w_tid.data(), (size_t)nwin * 4, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(d_w0, w_w0.data(), (size_t)nwin * 4, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(d_w1, w_w1.data(), (size_t)nwin * 4, hipMemcpyHostToDevice, st));
    A.pieces = d_p; A.wptr = d_wptr; A.w_tid = d_wt; A.w_w0 = d_w0; A.w_w1 = d_w1; A.nwin = (int32_t)nwin; A.min_depth = min_depth;
    A.q2 = queries->d_seq2; A.qn = queries->d_nmask; A.qboff = queries->d_boff; A.qlen = queries->d_len;
    A.t2 = tg->d_seq2; A.tn = tg->d_nmask; A.tboff = tg->d_boff;
    A.scratch = d_scr; A.slot_bytes = POA_SLOT_BYTES; A.wout = d_wout; A.wlen = d_wlen;
    hipLaunchKernelGGL(k_poa_window, dim3((unsigned)nslot), dim3(64), 0, st, A);
    CK(hipGetLastError());
    // the windows' strings leave the device packed (the slots of wout are POA_MAXNODE bytes each: 2 KB for ~200 bases)
    std::vector<int32_t> wlen((size_t)nwin);
    CK(hipMemcpyAsync(wlen.data(), d_wlen, (size_t)nwin * 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    std::vector<int64_t> woff((size_t)nwin + 1, 0);
    for (int64_t k = 0; k < nwin; ++k) woff[k + 1] = woff[k] + wlen[k];
    const int64_t tot = woff[nwin];
    int64_t *d_woff; uint8_t *d_pack;
    if ((rc = ctx_buf_t(ctx, "poa_woff", (size_t)nwin + 1, &d_woff)) != TELR_OK || (rc = ctx_buf_t(ctx, "poa_pack", (size_t)tot + 1, &d_pack)) != TELR_OK) return fail(rc);
    CK(hipMemcpyAsync(d_woff, woff.data(), ((size_t)nwin + 1) * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_poa_pack, dim3((unsigned)nwin), dim3(64), 0, st, d_wout, d_woff, d_pack);
    CK(hipGetLastError());
    C->seq.resize((size_t)tot);
    if (tot) CK(hipMemcpyAsync(&C->seq[0], d_pack, (size_t)tot, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
#undef CK
    for (int t = 0; t < nt; ++t) { C->off[t] = woff[wbase[t]]; C->len[t] = (int32_t)(woff[wbase[t + 1]] - woff[wbase[t]]); }
    *out = C;
    return TELR_OK;
}
extern "C" int telr_poa_build(telr_ctx *ctx, const telr_result *r, const telr_seqset *queries, const telr_index *idx, int32_t min_depth, telr_consensus **out)
{
    (void)hipGetLastError();
    int rc = poa_impl(ctx, r, queries, idx, min_depth, out);
    if (rc == TELR_E_NOMEM) {
        (void)hipGetLastError();
        mem_note(ctx, "telr_poa_build: out of memory");
        ctx_release_map_scratch(ctx, (uint64_t)256 * 32 * POA_SLOT_BYTES + ((uint64_t)1 << 30));
        rc = poa_impl(ctx, r, queries, idx, min_depth, out);
    }
    return rc;
}
