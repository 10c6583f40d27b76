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
//                   window borders it crosses -- in record order, CSR by window; the records are walked by the worker threads
//   k_poa_window    ONE WAVE per window, the windows handed out one by one (most pieces first) to as many waves as are resident:
//                   the graph lives in a slot of global scratch.  The score matrix is BANDED: a node's row holds the POA_BAND = 64
//                   columns around the diagonal of the node's window column (d_poa_lo; a piece is at most 30 bases off the diagonal
//                   by the POA_MAXINDEL rule), i.e. exactly one cell per lane.  Round 5 -- the kernel was a chain of dependent
//                   loads (SQ_WAIT_ANY 76 % of its wave cycles), 335 ms per 1,000 configs[2] loci; now 104 ms:
//                   * sweep: the per-rank tables (kind of row, base, band, first two predecessors) reach it 64 ranks at a time in
//                     registers (v_readlane); the set-up pass sorts the rows into kinds -- one predecessor = the row before (DPP
//                     lane shifts of the row held in registers), one or two predecessors within the last POA_RING rows (an LDS ring),
//                     anything else (the general loop; a predecessor further back than the ring comes from the slot, and only then
//                     are the rows written there); the in-row gap chain is a DPP prefix maximum; no lane masks (cells right of the
//                     piece's end are never read by cells left of them); a cell notes where its value came from (one byte)
//                   * walk back: by the whole wave -- a trip guesses up to 64 diagonal steps along first predecessors (the chain of
//                     rows followed in a register window of the predecessor table), loads the notes of the guessed cells and of the
//                     two columns either side at once, and follows the way through inserted bases / skipped nodes as long as it
//                     stays within them: 13 round trips to memory per piece instead of 800
//                   * merge of the piece into the graph and the new topological order: parallel passes over the path
//                   * heaviest bundle: 64 ranks at a time, the scores settled in rank order with v_readlane, the way back in a
//                     register window of the chosen-source table
//   k_poa_pack      the windows' strings, packed, for the copy back
// Integer work bounded by instruction issue (VALU and SALU alike) and L2 / MALL latency: no MFMA.  -DPOA_PROF: wave clocks per
// phase and row / trip counters on stderr (tools/poa_iter.sh).
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
#define POA_RING    16      /* rows of the sweep kept in LDS: a predecessor at most that many ranks back is read from there */
#define POA_SPEC    64      /* steps the walk back guesses per trip */

struct PoaPiece { int32_t qid, qa, len, rev; };           // query bases [qa, qa + len) on the alignment strand
struct PoaArgs {
    const PoaPiece *pieces; const int32_t *wptr;          // pieces of window w: [wptr[w], wptr[w + 1]) in record order
    const int32_t *w_tid, *w_w0, *w_w1; int32_t nwin, min_depth;
    const int32_t *wperm; int32_t *next;                  // the windows by falling number of pieces; the next one to hand out
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
#define POA_O_HLAST  (POA_O_LO + 2 * POA_MAXNODE)                            /* i16 [MAXNODE]: the last column of the row of rank r (-32000 outside its band) */
#define POA_O_H      ((POA_O_HLAST + 2 * POA_MAXNODE + 255) & ~255)            /* i16 [MAXNODE][POA_BAND]: the banded score matrix (row r = the node at rank r) */
#define POA_O_TB     (POA_O_H + 2 * POA_MAXNODE * POA_BAND)                 /* u8  [MAXNODE][POA_BAND]: where a cell's value came from (0-7: diagonal from predecessor k, 8-15: skipped the node, from predecessor k, 16: base inserted) */
#define POA_SLOT_BYTES ((size_t)POA_O_TB + (size_t)POA_MAXNODE * POA_BAND)

__host__ __device__ __forceinline__ int d_poa_lo(int col, int n, int L) { int lo = (col + 1) * n / L - POA_BAND / 2, hi = n + 1 - POA_BAND; if (lo > hi) lo = hi; return lo < 0 ? 0 : lo; }
// inclusive prefix maximum over the wave (DPP row shifts + the two row broadcasts; a lane without a source takes the identity: the compiler folds move and maximum into one v_max_i32_dpp)
__device__ __forceinline__ int d_wave_scan_max(int v)
{
    int t;
    t = __builtin_amdgcn_update_dpp((int)0x80000000, v, 0x111, 0xf, 0xf, false); v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp((int)0x80000000, v, 0x112, 0xf, 0xf, false); v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp((int)0x80000000, v, 0x114, 0xf, 0xf, false); v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp((int)0x80000000, v, 0x118, 0xf, 0xf, false); v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp((int)0x80000000, v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp((int)0x80000000, v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;
    return v;
}
// lane i reads lane i + 1 / lane i - 1 of the whole wave (a lane without a source gets `fill`)
__device__ __forceinline__ int d_wave_shl1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ int d_wave_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }

__device__ __forceinline__ int d_poa_qbase(const PoaArgs &A, const PoaPiece &P, int x)
{
    // base x of the piece on the alignment strand (a reverse record reads its query mirrored and complemented)
    const int64_t b0 = A.qboff[P.qid]; const int ql = A.qlen[P.qid];
    const int pos = P.rev ? ql - 1 - (P.qa + x) : P.qa + x;
    int b = d_base(A.q2, A.qn, b0 + pos);
    if (P.rev && b < 4) b = 3 - b;
    return b;
}

#ifdef POA_PROF
__device__ unsigned long long g_poa_prof[16];
#define POA_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); pf[i] += t_ - t_last; t_last = t_; } while (0)
#else
#define POA_T(i) do { } while (0)
#endif
__global__ void __launch_bounds__(64) k_poa_window(PoaArgs A)
{
#ifdef POA_PROF
    unsigned long long pf[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}, t_last = __builtin_readcyclecounter();
#define POA_C(i, x) pf[i] += (x)
#else
#define POA_C(i, x) do { } while (0)
#endif
    __shared__ uint8_t seq[POA_SEGMAX + 72];          // seq[x + 1] = base x of the piece; 7 (equal to no base) in front and for 64 entries behind
    __shared__ int32_t sel[POA_MAXSEG];
    __shared__ int32_t sh[8];
    __shared__ int16_t hring[POA_RING * POA_BAND];
    const int lane = threadIdx.x;
    uint8_t *S = A.scratch + (size_t)blockIdx.x * A.slot_bytes;
    uint8_t *base = S + POA_O_BASE, *nin = S + POA_O_NIN;
    int16_t *nout = (int16_t*)(S + POA_O_NOUT), *in = (int16_t*)(S + POA_O_IN), *inw = (int16_t*)(S + POA_O_INW), *ring = (int16_t*)(S + POA_O_RING);
    int16_t *order = (int16_t*)(S + POA_O_ORDER), *rank = (int16_t*)(S + POA_O_RANK), *startc = (int16_t*)(S + POA_O_STARTC), *endc = (int16_t*)(S + POA_O_ENDC);
    int16_t *bp = (int16_t*)(S + POA_O_BP), *no = (int16_t*)(S + POA_O_NO), *pn = (int16_t*)(S + POA_O_PN), *pj = (int16_t*)(S + POA_O_PJ);
    int16_t *newv = (int16_t*)(S + POA_O_NEWV), *anchor = (int16_t*)(S + POA_O_ANCH), *H = (int16_t*)(S + POA_O_H);
    int32_t *score = (int32_t*)(S + POA_O_SCORE);
    uint8_t *TB = S + POA_O_TB;
    int16_t *hlast = (int16_t*)(S + POA_O_HLAST);
    int16_t *prow = (int16_t*)(S + POA_O_PROW), *pplo = (int16_t*)(S + POA_O_PPLO), *pcb = (int16_t*)(S + POA_O_PCB), *col = (int16_t*)(S + POA_O_COL), *lo_r = (int16_t*)(S + POA_O_LO);
    // the windows are handed out one by one, the ones with most pieces first: a wave that drew short windows takes more of them
    // (dealt round-robin the busiest wave of 6,144 finished long after the average one)
    for (;;) {
        int wi = 0;
        if (lane == 0) wi = atomicAdd(A.next, 1);
        wi = __builtin_amdgcn_readfirstlane(wi);
        if (wi >= A.nwin) break;
        const int w = A.wperm[wi];
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
        POA_T(0);
        // ---- the graph starts as the draft's window
        int n = L;
        for (int v = lane; v < L; v += 64) {
            base[v] = (uint8_t)d_base(A.t2, A.tn, tb0 + w0 + v); nin[v] = v ? 1 : 0; nout[v] = v < L - 1 ? 1 : 0; ring[v] = (int16_t)v; order[v] = (int16_t)v;
            startc[v] = v == 0; endc[v] = v == L - 1; col[v] = (int16_t)v;
            if (v) { in[v * POA_MAXIN] = (int16_t)(v - 1); inw[v * POA_MAXIN] = 1; }
        }
        __syncthreads();
        POA_T(1);
        for (int si = 0; si < nsel; ++si) {
            const PoaPiece P = A.pieces[sel[si]];
            const int m = P.len;
            if (n + m > POA_MAXNODE) continue;
            for (int x = lane; x < m; x += 64) seq[x + 1] = (uint8_t)d_poa_qbase(A, P, x);
            seq[m + 1 + lane] = 7; if (lane == 0) seq[0] = 7;
            for (int r = lane; r < n; r += 64) { const int v = order[r]; rank[v] = (int16_t)(r + 1); lo_r[r] = (int16_t)d_poa_lo(col[v], m, L); }
            __syncthreads();
            // per RANK: the rows of the node's predecessors, the first columns of their bands and (count << 8 | base) -- the sweep reads
            // sequential tables, one node ahead, instead of chasing order -> in-list -> rank inside every step
            bool far_ = false;                           // a predecessor further back than the LDS ring: the rows go to the slot as well
            for (int r = lane; r < n; r += 64) {
                const int v = order[r], c = nin[v], np = c ? c : 1;
                bool ring_ = true; int pr0 = 0;
                for (int k = 0; k < np; ++k) {
                    const int pr = c ? rank[in[v * POA_MAXIN + k]] : 0;
                    prow[r * POA_MAXIN + k] = (int16_t)pr; pplo[r * POA_MAXIN + k] = pr ? lo_r[pr - 1] : 0;
                    far_ |= pr && r - (pr - 1) > POA_RING;
                    ring_ &= pr && r - (pr - 1) <= POA_RING;
                    if (k == 0) pr0 = pr;
                }
                // the kind of row the sweep meets: 0 / 1 = one predecessor, the row before, its band 0 / 1 columns to the left;
                // 2 = one or two predecessors within the ring; 3 = anything else
                const int dl = r ? lo_r[r] - lo_r[r - 1] : 9;
                const int kind = np == 1 && pr0 == r && (unsigned)dl <= 1u ? dl : np <= 2 && ring_ ? 2 : 3;
                // (the sweep's copy of the node's base: an ambiguous node -- an N of the draft -- is 6, which equals neither a piece's N (4) nor the pad (7):
                // the oracle scores N against N as a mismatch, tor_poa `seq[j-1] == base && seq[j-1] < 4`; the merge reads base[] itself)
                pcb[r] = (int16_t)((nout[v] ? 0 : 1) << 14 | kind << 12 | np << 8 | (base[v] < 4 ? base[v] : 6));      // bit 14: no out-edge -- a candidate for the end, its last column is noted
            }
            const bool need_h = __any(far_);
            __syncthreads();
            // ---- sweep: one row of POA_BAND cells per node, in topological order: lane l computes column lo + l.  Every cell also
            // notes where its value came from, by the walk's own preference (diagonal before skipped node before inserted base, the
            // first predecessor that explains it): the walk back then reads one byte per step instead of all candidates again.
            POA_T(2);
            // Round 5: the four per-rank tables reach the sweep 64 ranks at a time (lane i holds rank rb + i; a row reads them with
            // v_readlane), the batch after the running one already asked for: one load latency per 64 rows instead of one per row.
            // The in-row gap chain is a DPP prefix maximum, the row before is shifted by 0 / 1 / 2 lanes with DPP moves (no LDS
            // round trips), and the last POA_RING rows also stay in LDS: a predecessor a few ranks back -- the other arm of a
            // bubble -- is read from there; only a predecessor further back waits for the row stored in the slot.
            int c_cb = 0, c_lo = 0, n_cb = 0, n_lo = 0; uint32_t c_p01 = 0, c_l01 = 0, n_p01 = 0, n_l01 = 0;
            if (lane < n) { n_cb = pcb[lane]; n_lo = lo_r[lane]; n_p01 = *(const uint32_t*)&prow[lane * POA_MAXIN]; n_l01 = *(const uint32_t*)&pplo[lane * POA_MAXIN]; }
            int prev_val = -32000;
            // (two loops: the batch in flight must not be a value the ROW loop carries -- the compiler then copies it every row behind
            // an s_waitcnt vmcnt(0), which also waits for the row's stores: one store latency per row, 4,500 cycles measured)
            for (int rb = 0; rb < n; rb += 64) {
                c_cb = n_cb; c_lo = n_lo; c_p01 = n_p01; c_l01 = n_l01;
                { const int rr = rb + 64 + lane;
                  if (rr < n) { n_cb = pcb[rr]; n_lo = lo_r[rr]; n_p01 = *(const uint32_t*)&prow[rr * POA_MAXIN]; n_l01 = *(const uint32_t*)&pplo[rr * POA_MAXIN]; } }
                const int rcnt = n - rb < 64 ? n - rb : 64;
            for (int ri = 0; ri < rcnt; ++ri) {
                const int r = rb + ri;
                POA_C(9, 1);
                const int cb = __builtin_amdgcn_readlane(c_cb, ri), kind = (cb >> 12) & 3, np_ = (cb >> 8) & 15, vb = cb & 0xff, lo = __builtin_amdgcn_readlane(c_lo, ri), j = lo + lane;
                const int sc = seq[j] == vb ? POA_M : POA_X;
                int vmax = -1000000, vk = 0, dmax = -1000000, dk = 0;
                // Cells right of the piece's end (j > m) are computed like the others and never read by a cell left of them: a row reads
                // columns j and j - 1 of its predecessors, the gap chain runs left to right -- no lane masks in the row.
                if (kind < 2) {
                    // the common row, straight-line: one predecessor, the row before, its band 0 or 1 columns to the left
                    POA_C(10, 1);
                    const int s_r = d_wave_shr1(prev_val, -32000), s_l = d_wave_shl1(prev_val, -32000);
                    vmax = (kind ? s_l : prev_val) + POA_G;
                    dmax = (kind ? prev_val : s_r) + sc;
                } else {
                const uint32_t p01 = (uint32_t)__builtin_amdgcn_readlane((int)c_p01, ri), l01 = (uint32_t)__builtin_amdgcn_readlane((int)c_l01, ri);
                if (kind == 2) {
                    // one or two predecessors a few ranks back (the arms of a bubble): both from the ring in LDS, straight-line
                    POA_C(12, 1);
                    const bool two = np_ == 2;
                    const int pr0 = (int)(p01 & 0xffffu), pr1 = two ? (int)(p01 >> 16) : pr0, j0 = j - (int)(l01 & 0xffffu), j1 = j - (two ? (int)(l01 >> 16) : 20000);
                    const int16_t *rw0 = hring + ((pr0 - 1) & (POA_RING - 1)) * POA_BAND, *rw1 = hring + ((pr1 - 1) & (POA_RING - 1)) * POA_BAND;
                    const int a0 = rw0[j0 & (POA_BAND - 1)], b0 = rw0[(j0 - 1) & (POA_BAND - 1)], a1 = rw1[j1 & (POA_BAND - 1)], b1 = rw1[(j1 - 1) & (POA_BAND - 1)];
                    const int off1 = two ? -32000 : -1000000;
                    vmax = ((unsigned)j0 < (unsigned)POA_BAND ? a0 : -32000) + POA_G;
                    dmax = ((unsigned)(j0 - 1) < (unsigned)POA_BAND ? b0 : -32000) + sc;
                    const int c1 = ((unsigned)j1 < (unsigned)POA_BAND ? a1 : off1) + POA_G, d1 = ((unsigned)(j1 - 1) < (unsigned)POA_BAND ? b1 : off1) + sc;
                    if (c1 > vmax) { vmax = c1; vk = 1; }
                    if (d1 > dmax) { dmax = d1; dk = 1; }
                } else {
                bool synced = false;
                for (int k = 0; k < np_; ++k) {
                    const int pr = k == 0 ? (int)(p01 & 0xffffu) : k == 1 ? (int)(p01 >> 16) : __builtin_amdgcn_readfirstlane((int)prow[r * POA_MAXIN + k]);
                    const int pl = k == 0 ? (int)(l01 & 0xffffu) : k == 1 ? (int)(l01 >> 16) : __builtin_amdgcn_readfirstlane((int)pplo[r * POA_MAXIN + k]);
                    int vj, vj1;
                    if (pr == 0) { vj = j * POA_G; vj1 = (j - 1) * POA_G; }
                    else if (r - (pr - 1) <= POA_RING) {         // the row before or a few ranks back (the other arm of a bubble): the ring in LDS
                        const int jj = j - pl;
                        const int16_t *rw = hring + ((pr - 1) & (POA_RING - 1)) * POA_BAND;
                        const int a = rw[jj & (POA_BAND - 1)], b = rw[(jj - 1) & (POA_BAND - 1)];
                        vj = (jj >= 0 && jj < POA_BAND) ? a : -32000;
                        vj1 = (jj >= 1 && jj <= POA_BAND) ? b : -32000;
                    } else {
                        if (!synced) { __syncthreads(); synced = true; }        // (rows stored by other lanes of this wave)
                        POA_C(11, 1);
                        const int jj = j - pl;
                        const int16_t *prw = H + (size_t)(pr - 1) * POA_BAND;
                        vj = (jj >= 0 && jj < POA_BAND && j <= m) ? prw[jj] : -32000;
                        vj1 = (jj >= 1 && jj <= POA_BAND && j <= m) ? prw[jj - 1] : -32000;
                    }
                    int c = vj + POA_G; if (c > vmax) { vmax = c; vk = k; }
                    c = vj1 + sc; if (j > 0 && c > dmax) { dmax = c; dk = k; }
                }
                }
                }
                if (lo == 0) { if (lane == 0) dmax = -1000000; }      // column 0 has no diagonal (a uniform branch: most rows have none)
                int t = vmax > dmax ? vmax : dmax; t = t > -32000 ? t : -32000;
                // row[j] = max over the band's k <= j of T[k] + (j - k) G  =  (prefix max of T[k] - k G) + j G
                const int jg = j * -POA_G;
                const int cur = (int)(int16_t)(d_wave_scan_max(t + jg) - jg);
                if (need_h) H[(size_t)r * POA_BAND + lane] = (int16_t)cur;
                TB[(size_t)r * POA_BAND + (j & (POA_BAND - 1))] = (uint8_t)(dmax == cur ? dk : vmax == cur ? 8 | vk : 16);      // (the walk back finds a column's note without the row's band)
                prev_val = cur;
                if (cb >> 14) { if (lane == (m - lo < POA_BAND - 1 ? m - lo : POA_BAND - 1)) hlast[r] = (int16_t)(j == m ? cur : -32000); }      // (the end is chosen among the last columns of the nodes without out-edges)
                hring[(r & (POA_RING - 1)) * POA_BAND + lane] = (int16_t)cur;
            }
            }
            __syncthreads();
            POA_T(3);
            // ---- the end: the node without out-edges whose last column scores best, smallest id on ties
            {
                int bs = -32768, bv = 0x7fffffff;
                for (int v = lane; v < n; v += 64) if (!nout[v]) { const int sc = hlast[rank[v] - 1]; if (sc > bs) { bs = sc; bv = v; } }
#pragma unroll
                for (int s = 32; s >= 1; s >>= 1) { const int os = __shfl_xor(bs, s), ov = __shfl_xor(bv, s); if (os > bs || (os == bs && ov < bv)) { bs = os; bv = ov; } }
                if (lane == 0) sh[0] = bv == 0x7fffffff ? -1 : bv;
            }
            __syncthreads();
            POA_T(4);
            {
                // ---- walk back along the notes of the sweep (pn holds ROW numbers here, 0 = no node; turned into nodes below).
                // Round 5: the whole wave walks.  One step at a time from lane 0 was two dependent loads from the slot per step (the
                // note, then the predecessor's row): ~800 round trips to memory per piece, most of the kernel.  Now a trip GUESSES the
                // next POA_SPEC steps -- diagonal moves along first predecessors, the common run between two differences -- lane l
                // taking step l: the rows come from a window of the first-predecessor table held in registers (64 ranks, followed with
                // v_readlane), the notes of all guessed cells are loaded at once, and the steps up to the first note that is not
                // "diagonal from the first predecessor" are emitted together; that one step is then taken as noted.
                int np = 0, j = m;
                int r = __builtin_amdgcn_readfirstlane(sh[0] >= 0 ? (int)rank[sh[0]] : 0);
                int wbase = -1, nbase = -1; uint32_t pp = 0, pnx = 0;     // lane i: rows of the first two predecessors of row wbase - i; the window asked for ahead
                for (int it = 0; (r > 0 || j > 0) && j >= 0 && it < 2 * POA_NPATH; ++it) {        // (the bounds only keep a broken slot from hanging the device)
                    if (r == 0) {                                  // bases before the graph's start
                        for (int x = lane; x < j; x += 64) { pn[np + x] = 0; pj[np + x] = (int16_t)(j - 1 - x); }
                        np += j; j = 0; break;
                    }
                    POA_C(13, 1);
                    if (wbase < r || wbase - r > 63) {
                        if (nbase >= r && nbase - r <= 63) { pp = pnx; wbase = nbase; nbase = -1; }
                        else { POA_C(14, 1); wbase = r; const int rr = r - lane; pp = rr >= 1 ? *(const uint32_t*)&prow[(rr - 1) * POA_MAXIN] : 0u; }
                    }
                    // the chain of first predecessors from r, step l on lane l: rows whose first predecessor is the row before them are
                    // taken a run at a time (a ballot over the window), the others one v_readlane each
                    const uint64_t consec = __ballot(wbase - lane >= 1 && (int)(pp & 0xffffu) == wbase - lane - 1);
                    int mine = 0, c = r, len = 0;
                    while (len < POA_SPEC && c != 0 && wbase - c <= 63 && len <= j) {
                        const int ci = wbase - c;
                        const uint64_t rest = ~(consec >> ci);
                        const int full = (rest ? (int)__builtin_ctzll(rest) : 64) + 1;    // the run and the row that ends it
                        int take = full;
                        if (take > 64 - ci) take = 64 - ci;                               // (the window's end, the rows above the virtual start,
                        if (take > c) take = c;                                           //  the lanes, the columns left)
                        if (take > POA_SPEC - len) take = POA_SPEC - len;
                        if (take > j - len + 1) take = j - len + 1;
                        const bool whole = take == full;
                        if (lane >= len && lane < len + take) mine = c - (lane - len);
                        c = whole ? (int)((uint32_t)__builtin_amdgcn_readlane((int)pp, ci + take - 1) & 0xffffu) : c - take;
                        len += take;
                    }
                    // the guess left the window: the window it ends in is asked for together with the notes (one round trip for both)
                    if (c > 0 && wbase - c > 63) { nbase = c; const int rr = c - lane; pnx = rr >= 1 ? *(const uint32_t*)&prow[(rr - 1) * POA_MAXIN] : 0u; }
                    // the notes of the guessed cells AND of the cells up to two columns left / right of them: an inserted base moves the
                    // way one column left of the guess, a skipped node one column right -- the trip goes on through such steps as long as
                    // the way stays within two columns of the guess (five 5-bit notes per lane in one register)
                    const int jl = j - lane;
                    uint32_t pk = 0x1084210u;                     // 16 (no note: an inserted base) five times
                    if (lane < len) {
                        const int lo_l = lo_r[mine - 1];
                        const uint8_t *tr = TB + (size_t)(mine - 1) * POA_BAND;
                        uint32_t q = 0;
#pragma unroll
                        for (int d = -2; d <= 2; ++d) {
                            const int col = jl + d, jj = col - lo_l;
                            const uint32_t t = tr[col & (POA_BAND - 1)];
                            q |= (col >= 0 && jj >= 0 && jj < POA_BAND ? t : 16u) << (5 * (d + 2));
                        }
                        pk = q;
                    }
                    uint64_t okm[5];
#pragma unroll
                    for (int d = 0; d < 5; ++d) okm[d] = __ballot(lane < len && ((pk >> (5 * d)) & 31u) == 0u);
                    int l = 0, d = 0;                              // the way is at the chain's step l, d columns right of the guess
                    int nr = -1, nj = 0;                           // the state the next trip starts from
                    for (int ev = 0; ev < 2 * POA_SPEC + 8; ++ev) {
                        // plain diagonal steps from here
                        const uint64_t om = d == -2 ? okm[0] : d == -1 ? okm[1] : d == 0 ? okm[2] : d == 1 ? okm[3] : okm[4];
                        const uint64_t rest = ~(om >> l);
                        int run = rest ? (int)__builtin_ctzll(rest) : 64; if (run > len - l) run = len - l;
                        if (lane >= l && lane < l + run) { pn[np + lane - l] = (int16_t)mine; pj[np + lane - l] = (int16_t)(jl + d - 1); }
                        np += run; l += run;
                        if (l >= len) { nr = c; nj = j - l + d; break; }
                        const int col = j - l + d;
                        const int tbs = (int)(((uint32_t)__builtin_amdgcn_readlane((int)pk, l) >> (5 * (d + 2))) & 31u);
                        if (tbs == 8) {                            // the node is skipped (first predecessor): the chain's next row, the same column
                            ++l; ++d;
                            if (l >= len) { nr = c; nj = col; break; }
                            if (d > 2) { nr = __builtin_amdgcn_readlane(mine, l); nj = col; break; }
                        } else if (tbs == 16) {                    // an inserted base: the same row, one column left
                            if (lane == 0) { pn[np] = 0; pj[np] = (int16_t)(col - 1); }
                            ++np; --d;
                            if (d < -2 || col - 1 < 0) { nr = __builtin_amdgcn_readlane(mine, l); nj = col - 1; break; }
                        } else {                                   // another predecessor: the trip ends with that step
                            const int rr = __builtin_amdgcn_readlane(mine, l), k = tbs & 7;
                            if (tbs < 8) { if (lane == 0) { pn[np] = (int16_t)rr; pj[np] = (int16_t)(col - 1); } ++np; }
                            const uint32_t p01 = (uint32_t)__builtin_amdgcn_readlane((int)pp, wbase - rr);
                            nr = k == 0 ? (int)(p01 & 0xffffu) : k == 1 ? (int)(p01 >> 16) : __builtin_amdgcn_readfirstlane((int)prow[(rr - 1) * POA_MAXIN + k]);
                            nj = tbs < 8 ? col - 1 : col;
                            break;
                        }
                    }
                    if (nr < 0) { nr = __builtin_amdgcn_readlane(mine, l); nj = j - l + d; }      // (the bound of the loop above: not reached)
                    r = nr; j = nj;
                }
                if (lane == 0) sh[3] = np;
            }
            __syncthreads();
            for (int z = lane; z < sh[3]; z += 64) { const int r = pn[z]; pn[z] = r ? order[r - 1] : (int16_t)-1; }
            __syncthreads();
            POA_T(5);
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
                    if (on) { x = pn[z]; b = seq[pj[z] + 1]; }
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
                POA_T(6);
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
            POA_T(7);
        }
        POA_T(7);
        // ---- heaviest bundle between the most common start / end nodes.  Round 5: one step at a time from lane 0 this was four
        // dependent loads per node and one per step of the way back (2 ms of a window's 10).  Now 64 ranks at a time: every lane
        // loads its node's in-edges and the ranks / scores of their sources at once; the scores are then settled in rank order
        // with v_readlane -- a source in the same 64 ranks is read from its lane's register, an earlier one came with
        // the loads -- and stored by all lanes together; bp[] holds the RANK of the chosen source, so the way back is followed in a
        // register window as well.
        for (int r = lane; r < n; r += 64) rank[order[r]] = (int16_t)(r + 1);
        __syncthreads();
        for (int r0 = 0; r0 < n; r0 += 64) {
            const int rr = r0 + lane, cnt = n - r0 < 64 ? n - r0 : 64;
            int v = 0, c = 0; uint32_t i01 = 0, i23 = 0, w01 = 0, w23 = 0, r01 = 0, r23 = 0; int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            if (rr < n) {
                v = order[rr]; c = nin[v];
                const uint2 iv = *(const uint2*)&in[v * POA_MAXIN], wv = *(const uint2*)&inw[v * POA_MAXIN];
                i01 = iv.x; i23 = iv.y; w01 = wv.x; w23 = wv.y;
                if (c > 0) { const int u = (int)(i01 & 0xffffu); r01 = (uint32_t)(uint16_t)rank[u]; s0 = score[u]; }
                if (c > 1) { const int u = (int)(i01 >> 16); r01 |= (uint32_t)(uint16_t)rank[u] << 16; s1 = score[u]; }
                if (c > 2) { const int u = (int)(i23 & 0xffffu); r23 = (uint32_t)(uint16_t)rank[u]; s2 = score[u]; }
                if (c > 3) { const int u = (int)(i23 >> 16); r23 |= (uint32_t)(uint16_t)rank[u] << 16; s3 = score[u]; }
            }
            int sc = 0, bpr = 0;
            for (int l = 0; l < cnt; ++l) {
                const int cl_ = __builtin_amdgcn_readlane(c, l);
                int bw = -1, bs = -1, br = 0;
                for (int k = 0; k < cl_; ++k) {
                    int wgt, ru, su;
                    if (k < 4) {
                        const uint32_t wp = (uint32_t)__builtin_amdgcn_readlane((int)(k < 2 ? w01 : w23), l), rp = (uint32_t)__builtin_amdgcn_readlane((int)(k < 2 ? r01 : r23), l);
                        wgt = (int)(int16_t)((k & 1) ? wp >> 16 : wp & 0xffffu); ru = (int)((k & 1) ? rp >> 16 : rp & 0xffffu);
                        su = __builtin_amdgcn_readlane(k == 0 ? s0 : k == 1 ? s1 : k == 2 ? s2 : s3, l);
                    } else {
                        const int vv = __builtin_amdgcn_readlane(v, l);
                        const int u = __builtin_amdgcn_readfirstlane((int)in[vv * POA_MAXIN + k]);
                        wgt = __builtin_amdgcn_readfirstlane((int)inw[vv * POA_MAXIN + k]); ru = __builtin_amdgcn_readfirstlane((int)rank[u]); su = __builtin_amdgcn_readfirstlane(score[u]);
                    }
                    if (ru - 1 >= r0) su = __builtin_amdgcn_readlane(sc, ru - 1 - r0);      // settled in this pass: not in memory yet
                    if (wgt > bw || (wgt == bw && su > bs)) { bw = wgt; bs = su; br = ru; }
                }
                if (lane == l) { sc = cl_ > 0 ? bs + bw : 0; bpr = br; }
            }
            if (rr < n) { score[v] = sc; bp[rr] = (int16_t)bpr; }
            __syncthreads();
        }
        {
            // the end: most sequences end there, better score, smaller id; the start: most sequences begin there, smaller id
            int ec = -1, es = -1, ev = 0x7fffffff, stc = -1, stv = 0x7fffffff;
            for (int v = lane; v < n; v += 64) {
                const int e = endc[v], sc_ = score[v], st = startc[v];
                if (e > ec || (e == ec && sc_ > es)) { ec = e; es = sc_; ev = v; }
                if (st > stc) { stc = st; stv = v; }
            }
#pragma unroll
            for (int x = 32; x >= 1; x >>= 1) {
                const int oc = __shfl_xor(ec, x), os = __shfl_xor(es, x), ov = __shfl_xor(ev, x), otc = __shfl_xor(stc, x), otv = __shfl_xor(stv, x);
                if (oc > ec || (oc == ec && (os > es || (os == es && ov < ev)))) { ec = oc; es = os; ev = ov; }
                if (otc > stc || (otc == stc && otv < stv)) { stc = otc; stv = otv; }
            }
            // the way back over ranks (1-based; bp[rank - 1] = rank of the chosen source, 0 = none)
            int cur = __builtin_amdgcn_readfirstlane((int)rank[ev]); const int rs = __builtin_amdgcn_readfirstlane((int)rank[stv]);
            int cl = 0, wb = -1, bw_ = 0;                          // lane i: bp of rank wb - i
            for (int it = 0; cur > 0 && it < POA_MAXNODE; ++it) {
                if (wb < 0 || wb - cur > 63) { wb = cur; const int q_ = cur - lane; bw_ = q_ >= 1 ? (int)bp[q_ - 1] : 0; }
                if (lane == 0) no[cl] = (int16_t)cur;
                ++cl;
                if (cur == rs) break;
                cur = __builtin_amdgcn_readlane(bw_, wb - cur);
            }
            __syncthreads();
            for (int i = lane; i < cl; i += 64) out[i] = "ACGTN"[base[order[no[cl - 1 - i] - 1]]];
            if (lane == 0) A.wlen[w] = cl;
        }
        __syncthreads();
        POA_T(8);
    }
#ifdef POA_PROF
    if (lane == 0) for (int i = 0; i < 16; ++i) atomicAdd(&g_poa_prof[i], pf[i]);
#endif
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
    HostTrace ht("poa");
    result_wait(r);
    ht.mark("result ready");
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
    // (round 5: the records are walked by the host's worker threads, ranges of equal CIGAR length, every thread's pieces kept in
    // record order -- one thread took 128 ms per 1,000 loci, more than the kernel)
    const int64_t nal = (int64_t)r->alns.size();
    const int NTH = std::max(1, std::min(host_threads(), 32));
    std::vector<int64_t> cut((size_t)NTH + 1, nal);
    {   int64_t tot_ops = 0; for (const telr_aln &a : r->alns) tot_ops += a.n_cigar + 16;
        int64_t acc = 0; int t = 0; cut[0] = 0;
        for (int64_t x = 0; x < nal && t + 1 < NTH; ++x) { acc += r->alns[(size_t)x].n_cigar + 16; while (t + 1 < NTH && acc >= tot_ops * (t + 1) / NTH) cut[(size_t)++t] = x + 1; }
    }
    std::vector<std::vector<Cand>> tc((size_t)NTH);
    HostPool::get().run(NTH, [&](int t) {
      {
        std::vector<Cand> &cand = tc[(size_t)t];
        std::vector<int32_t> bval; std::vector<uint8_t> bbig;
        for (int64_t x_ = cut[(size_t)t]; x_ < cut[(size_t)t + 1]; ++x_) {
        const telr_aln &a = r->alns[(size_t)x_];
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
      }
    });
    { size_t tot_c = 0; for (const auto &v : tc) tot_c += v.size(); cand.reserve(tot_c); for (const auto &v : tc) cand.insert(cand.end(), v.begin(), v.end()); }
    ht.mark("pieces from the CIGARs");
    std::vector<int32_t> wptr((size_t)nwin + 1, 0);
    for (const Cand &c : cand) ++wptr[c.win + 1];
    for (int64_t k = 0; k < nwin; ++k) wptr[k + 1] += wptr[k];
    std::vector<PoaPiece> pieces(cand.size());
    { std::vector<int32_t> fill(wptr.begin(), wptr.end() - 1); for (const Cand &c : cand) pieces[fill[c.win]++] = c.p; }
    std::vector<int32_t> wperm((size_t)nwin);
    {   // counting sort by the number of pieces, most first (ties: window order)
        int32_t mx = 0; for (int64_t k = 0; k < nwin; ++k) mx = std::max(mx, wptr[k + 1] - wptr[k]);
        std::vector<int64_t> cnt((size_t)mx + 2, 0);
        for (int64_t k = 0; k < nwin; ++k) ++cnt[(size_t)(mx - (wptr[k + 1] - wptr[k])) + 1];
        for (int32_t c = 0; c <= mx; ++c) cnt[(size_t)c + 1] += cnt[c];
        for (int64_t k = 0; k < nwin; ++k) wperm[(size_t)cnt[(size_t)(mx - (wptr[k + 1] - wptr[k]))]++] = (int32_t)k;
    }
    // one slot per RESIDENT wave (a wave beyond that would start when the others have done all their windows: twice the time)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_poa_window, 64, 0) != hipSuccess || per_cu < 1) per_cu = 8;
    hipDeviceProp_t prop; int ncu = 256;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    const int64_t want = (int64_t)per_cu * ncu;
    const int nslot = (int)std::min<int64_t>(nwin, want);
    PoaArgs A; memset(&A, 0, sizeof(A));
    PoaPiece *d_p; int32_t *d_wptr, *d_wt, *d_w0, *d_w1, *d_wlen, *d_wperm, *d_next; uint8_t *d_scr, *d_wout;
    int rc;
    auto fail = [&](int code) { delete C; return code; };
    if ((rc = ctx_buf_t(ctx, "poa_pieces", pieces.size() + 1, &d_p)) != TELR_OK || (rc = ctx_buf_t(ctx, "poa_wptr", (size_t)nwin + 1, &d_wptr)) != TELR_OK ||
        (rc = ctx_buf_t(ctx, "poa_wt", (size_t)nwin, &d_wt)) != TELR_OK || (rc = ctx_buf_t(ctx, "poa_w0", (size_t)nwin, &d_w0)) != TELR_OK ||
        (rc = ctx_buf_t(ctx, "poa_w1", (size_t)nwin, &d_w1)) != TELR_OK || (rc = ctx_buf_t(ctx, "poa_wlen", (size_t)nwin, &d_wlen)) != TELR_OK ||
        (rc = ctx_buf_t(ctx, "poa_wperm", (size_t)nwin, &d_wperm)) != TELR_OK || (rc = ctx_buf_t(ctx, "poa_next", (size_t)4, &d_next)) != TELR_OK ||
        (rc = ctx_buf_t(ctx, "poa_scratch", (size_t)nslot * POA_SLOT_BYTES, &d_scr)) != TELR_OK ||
        (rc = ctx_buf_t(ctx, "poa_wout", (size_t)nwin * POA_MAXNODE, &d_wout)) != TELR_OK) return fail(rc);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ctx->err = std::string(#x) + ": " + hipGetErrorString(e_); return fail(e_ == hipErrorOutOfMemory ? TELR_E_NOMEM : TELR_E_HIP); } } while (0)
    if (!pieces.empty()) CK(hipMemcpyAsync(d_p, pieces.data(), pieces.size() * sizeof(PoaPiece), hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(d_wptr, wptr.data(), ((size_t)nwin + 1) * 4, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(d_wt, w_tid.data(), (size_t)nwin * 4, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(d_w0, w_w0.data(), (size_t)nwin * 4, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(d_w1, w_w1.data(), (size_t)nwin * 4, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(d_wperm, wperm.data(), (size_t)nwin * 4, hipMemcpyHostToDevice, st));
    CK(hipMemsetAsync(d_next, 0, 4, st));
    A.pieces = d_p; A.wptr = d_wptr; A.w_tid = d_wt; A.w_w0 = d_w0; A.w_w1 = d_w1; A.nwin = (int32_t)nwin; A.min_depth = min_depth; A.wperm = d_wperm; A.next = d_next;
    A.q2 = queries->d_seq2; A.qn = queries->d_nmask; A.qboff = queries->d_boff; A.qlen = queries->d_len;
    A.t2 = tg->d_seq2; A.tn = tg->d_nmask; A.tboff = tg->d_boff;
    A.scratch = d_scr; A.slot_bytes = POA_SLOT_BYTES; A.wout = d_wout; A.wlen = d_wlen;
    ht.mark("grouped, uploaded");
    hipLaunchKernelGGL(k_poa_window, dim3((unsigned)nslot), dim3(64), 0, st, A);
    CK(hipGetLastError());
    // the windows' strings leave the device packed (the slots of wout are POA_MAXNODE bytes each: 2 KB for ~200 bases)
    std::vector<int32_t> wlen((size_t)nwin);
    CK(hipMemcpyAsync(wlen.data(), d_wlen, (size_t)nwin * 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    ht.mark("k_poa_window");
    std::vector<int64_t> woff((size_t)nwin + 1, 0);
    for (int64_t k = 0; k < nwin; ++k) woff[k + 1] = woff[k] + wlen[k];
    const int64_t tot = woff[nwin];
    int64_t *d_woff; uint8_t *d_pack;
    if ((rc = ctx_buf_t(ctx, "poa_woff", (size_t)nwin + 1, &d_woff)) != TELR_OK || (rc = ctx_buf_t(ctx, "poa_pack", (size_t)tot + 1, &d_pack)) != TELR_OK) return fail(rc);
    CK(hipMemcpyAsync(d_woff, woff.data(), ((size_t)nwin + 1) * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_poa_pack, dim3((unsigned)nwin), dim3(64), 0, st, d_wout, d_woff, d_pack);
    CK(hipGetLastError());
#ifdef POA_PROF
    { unsigned long long h[16]; CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_poa_prof), sizeof(h))); unsigned long long z[16] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_poa_prof), z, sizeof(z)));
      double tot_ = 0; for (int i = 0; i < 9; ++i) tot_ += (double)h[i];
      static const char *nm[9] = {"select", "init", "setup", "sweep", "end", "walk", "merge", "order", "bundle"};
      fprintf(stderr, "[poa prof] windows %lld pieces %zu:", (long long)nwin, pieces.size()); for (int i = 0; i < 9; ++i) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * (double)h[i] / (tot_ > 0 ? tot_ : 1)); fprintf(stderr, "\n");
      fprintf(stderr, "[poa prof] rows %llu fast %.1f%% ring reads %.1f%% slot reads %.2f%% of rows; walk trips %.1f per piece, window loads %.1f per piece\n", h[9], 100.0 * h[10] / (h[9] ? h[9] : 1), 100.0 * h[12] / (h[9] ? h[9] : 1), 100.0 * h[11] / (h[9] ? h[9] : 1), (double)h[13] / (pieces.size() ? pieces.size() : 1), (double)h[14] / (pieces.size() ? pieces.size() : 1)); }
#endif
    C->seq.resize((size_t)tot);
    if (tot) CK(hipMemcpyAsync(&C->seq[0], d_pack, (size_t)tot, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
#undef CK
    ht.mark("packed, copied back");
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
