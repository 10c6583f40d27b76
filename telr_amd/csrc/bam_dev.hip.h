// bam_dev.hip.h — the stage-1 hand-off H1 built on the device: coordinate-sorted BAM + .bai straight from the resident
// reads, reference and CIGARs (replaces `samtools sort -o BAM SAM; samtools index BAM`, reference
// src/telr/TELR_alignment.py:103-114, and the SAM text minimap2 / ngmlr would have written first, :28-82).
//
// Included at the end of telr_engine.hip (same translation unit: it uses the context's grow-only buffers).
//
//   host: records of the result (+ one pseudo record per unmapped read), CIGAR array, names  -> HBM
//   k_bam_scan   one wave per record: walks the CIGAR against the 2-bit reads / reference, 64 ops per trip
//                -> NM, lengths of the MD and cs strings, inserted / deleted bases (for SA)
//   k_bam_size   one thread per record: SA length, record size, sort key (refID, pos, strand)
//   rocPRIM      stable radix sort of the keys, scan of the sizes in sorted order -> offset of every record
//   k_bam_write  one wave per record: the whole BAM record (fixed part, name, CIGAR, 4-bit SEQ, QUAL 0xff, tags with
//                MD / cs / SA text) at its place of the uncompressed stream
//   k_bgzf_*     one workgroup per 65280-byte block of that stream: CRC-32 + framing (level 0: stored blocks;
//                level >= 1: Huffman-coded deflate blocks, see below)
//   host: DMA of the finished file image in chunks through a pinned ring into the file, .bai from the record offsets
//
// Byte-level work bounded by HBM and then by PCIe / the file system: no matrix cores anywhere.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <deque>

#define BAM_BLK 65280            /* uncompressed bytes per BGZF block (as the host writer) */

struct BamInfo { int32_t nm, md_len, cs_len, nI, nD; };
struct BamArgs {
    const telr_aln *alns; int32_t nrec, n_mapped;
    const uint32_t *cig;
    const uint32_t *q2, *qn; const int64_t *qboff;
    const uint32_t *t2, *tn; const int64_t *tboff;
    const char *qnames; const int64_t *qname_off;      // names with their NUL; [nq + 1]
    const char *tnames; const int32_t *tname_off;      // [nt + 1]
    const char *rg; int32_t rg_len;                     // 0 = no RG tag
    int32_t flags;                                      // TELR_SAM_*
    BamInfo *info; uint32_t *rec_size; uint64_t *key; const uint64_t *rec_ustart; uint8_t *ubuf;
    const uint8_t *emit;                                // slice mode (nullable): record k is written iff emit[k]; the others are only there for the SA tags of their read
    const uint32_t *order; int32_t s0;                  // k_bam_write in pieces of the SORTED order: block x writes record order[s0 + x] (order null: record x)
};

typedef uint32_t __attribute__((aligned(1))) u32_unal;
__device__ __forceinline__ void d_st32(uint8_t *p, uint32_t v) { *(u32_unal*)p = v; }
__device__ __forceinline__ void d_st16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
__device__ __forceinline__ int d_ndig(uint32_t v)
{
    return 1 + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) + (v >= 100000000u) + (v >= 1000000000u);
}
__device__ __forceinline__ uint8_t *d_put_uint(uint8_t *p, uint32_t v)
{
    const int nd = d_ndig(v);
    for (int i = nd - 1; i >= 0; --i) { p[i] = (uint8_t)('0' + v % 10u); v /= 10u; }
    return p + nd;
}
// a lane's piece of tag text, written four bytes at a time (a quarter of the store instructions; measured: no change in the
// kernel's 29 ms for a 30x set, which is not bound by them)
struct ByteW {
    uint8_t *p; uint32_t acc; int k;
    __device__ __forceinline__ void init(uint8_t *q) { p = q; acc = 0; k = 0; }
    __device__ __forceinline__ void put(uint32_t b) { acc |= b << (8 * k); if (++k == 4) { d_st32(p, acc); p += 4; acc = 0; k = 0; } }
    __device__ __forceinline__ void put_uint(uint32_t v)
    {
        uint64_t bcd = 0; int nd = 0;           // digits, last one first; read back from the low end: first one first
        do { bcd = bcd << 4 | (uint64_t)(v % 10u); v /= 10u; ++nd; } while (v);
        for (int i = 0; i < nd; ++i) { put('0' + (uint32_t)(bcd & 15u)); bcd >>= 4; }
    }
    __device__ __forceinline__ void flush() { for (int i = 0; i < k; ++i) p[i] = (uint8_t)(acc >> (8 * i)); k = 0; acc = 0; }
};
__device__ __forceinline__ int d_wave_sum(int v) { for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s); return v; }
// inclusive prefix sum over the wave
__device__ __forceinline__ int d_wave_incl(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(v, o); if (lane >= o) v += y; }
    return v;
}

// 32 bases x .. x+31 of a sequence (first base at base offset b0 of the packed arrays, L bases) read on strand `rev`
// (1: the reverse complement, x counted from its own start), first base in the low bits; n = ambiguity bits.
// Positions beyond the sequence hold garbage.
struct B32 { uint64_t w; uint32_t n; };
__device__ __forceinline__ B32 d_fetch32(const uint32_t *__restrict__ s2, const uint32_t *__restrict__ nm, int64_t b)
{
    B32 r;
    const int64_t w = b >> 4; const int sh = (int)(b & 15) * 2;
    const uint64_t lo = (uint64_t)s2[w] | (uint64_t)s2[w + 1] << 32, hi = s2[w + 2];
    r.w = sh ? (lo >> sh | hi << (64 - sh)) : lo;
    const int64_t wn = b >> 5; const int shn = (int)(b & 31);
    r.n = (uint32_t)(((uint64_t)nm[wn] | (uint64_t)nm[wn + 1] << 32) >> shn);
    return r;
}
__device__ __forceinline__ B32 d_strand32(const uint32_t *__restrict__ s2, const uint32_t *__restrict__ nm, int64_t b0, int L, int rev, int x)
{
    if (!rev) return d_fetch32(s2, nm, b0 + x);
    int f0 = L - 32 - x;
    const int pad = f0 < 0 ? -f0 : 0;
    if (f0 < 0) f0 = 0;
    B32 f = d_fetch32(s2, nm, b0 + f0);
    if (pad) { f.w = pad < 32 ? f.w << (2 * pad) : 0; f.n = pad < 32 ? f.n << pad : 0; }
    B32 r;
    const uint64_t t = __brevll(f.w);
    r.w = ~(((t & 0x5555555555555555ULL) << 1) | ((t >> 1) & 0x5555555555555555ULL));
    r.n = __brev(f.n);
    return r;
}
// bit j set = columns j of the two 32-base words differ (or either base is ambiguous); limited to `n` columns
__device__ __forceinline__ uint32_t d_mis32(const B32 &q, const B32 &t, int n)
{
    uint64_t x = q.w ^ t.w;
    x = (x | x >> 1) & 0x5555555555555555ULL;
    x = (x | x >> 1) & 0x3333333333333333ULL;
    x = (x | x >> 2) & 0x0f0f0f0f0f0f0f0fULL;
    x = (x | x >> 4) & 0x00ff00ff00ff00ffULL;
    x = (x | x >> 8) & 0x0000ffff0000ffffULL;
    x = (x | x >> 16) & 0xffffffffULL;
    const uint32_t m = (uint32_t)x | q.n | t.n;
    return n >= 32 ? m : m & ((1u << n) - 1u);
}
__device__ __forceinline__ uint8_t d_base_char(const B32 &b, int j) { return (b.n >> j & 1u) ? (uint8_t)'N' : (uint8_t)("ACGT"[(b.w >> (2 * j)) & 3u]); }

// 8 bases (2-bit codes in the low 16 bits of c, ambiguity bits in the low 8 bits of n) -> 4 bytes of BAM 4-bit codes
// (A 1, C 2, G 4, T 8, N 15; first base of a pair in the HIGH nibble)
__device__ __forceinline__ uint32_t d_nib8(uint32_t c, uint32_t n)
{
    uint32_t x = c & 0xffffu;
    x = (x | x << 8) & 0x00ff00ffu; x = (x | x << 4) & 0x0f0f0f0fu; x = (x | x << 2) & 0x33333333u;
    const uint32_t b0 = x & 0x11111111u, b1 = (x >> 1) & 0x11111111u, n0 = b0 ^ 0x11111111u, n1 = b1 ^ 0x11111111u;
    uint32_t r = (n1 & n0) | (n1 & b0) << 1 | (b1 & n0) << 2 | (b1 & b0) << 3;
    uint32_t m = n & 0xffu;
    m = (m | m << 12) & 0x000f000fu; m = (m | m << 6) & 0x03030303u; m = (m | m << 3) & 0x11111111u;
    r |= m * 15u;
    return (r & 0x0f0f0f0fu) << 4 | (r >> 4 & 0x0f0f0f0fu);
}

__device__ __forceinline__ int d_reg2bin(int beg, int end)
{
    --end;
    if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (beg >> 14);
    if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (beg >> 17);
    if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (beg >> 20);
    if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (beg >> 23);
    if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (beg >> 26);
    return 0;
}

// ---- the CIGAR walk of one record by one wave: 64 ops per trip ---------------------------------------------------------
// MD: numbers count the matching columns since the last mismatch / deletion and run on across insertions and op
// boundaries, so the count a lane starts with is a segmented sum over the lanes before it (and the trips before this one).
// cs (short form): every op is self-contained.  WRITE = false: lengths only.
template <bool WRITE>
__device__ __forceinline__ void d_bam_walk(const BamArgs &A, const telr_aln &a, int lane, uint8_t *md_p, uint8_t *cs_p, BamInfo &out)
{
    const int rev = (a.flags & TELR_F_REV) ? 1 : 0;
    const int ql = a.qlen;
    const int64_t qb0 = A.qboff[a.qid], tb0 = A.tboff[a.tid];
    const uint32_t *__restrict__ cg = A.cig + a.cigar_off;
    const bool want_md = (A.flags & TELR_SAM_MD) != 0, want_cs = (A.flags & TELR_SAM_CS) != 0;
    int qi0 = rev ? ql - a.qe : a.qs, ti0 = a.ts;
    uint32_t carry = 0;
    int nm = 0, nI = 0, nD = 0, md_sum = 0, cs_sum = 0;
    uint32_t md_base = 0, cs_base = 0;
    for (int z0 = 0; z0 < a.n_cigar; z0 += 64) {
        const int z = z0 + lane;
        const bool have = z < a.n_cigar;
        const uint32_t c = have ? cg[z] : 0u;
        const int op = (int)(c & 0xfu), L = (int)(c >> 4);
        const int qadv = (have && op != 2) ? L : 0, tadv = (have && op != 1) ? L : 0;
        const int qinc = d_wave_incl(qadv, lane), tinc = d_wave_incl(tadv, lane);
        const int qi = qi0 + qinc - qadv, ti = ti0 + tinc - tadv;
        // ---- what this op contributes
        bool ev = false; uint32_t lead = 0, trail = 0, tot = 0; int md_rest = 0, cs_b = 0;
        B32 qw0, tw0; qw0.w = tw0.w = 0; qw0.n = tw0.n = 0;          // the op's first 32 columns: most M ops are shorter, and the write pass below reads them again
        if (have) {
            if (op == 0) {
                int first = -1, last = -1;
                for (int p0 = 0; p0 < L; p0 += 32) {
                    const B32 qw = d_strand32(A.q2, A.qn, qb0, ql, rev, qi + p0), tw = d_fetch32(A.t2, A.tn, tb0 + ti + p0);
                    if (WRITE && p0 == 0) { qw0 = qw; tw0 = tw; }
                    uint32_t mis = d_mis32(qw, tw, L - p0);
                    nm += __popc(mis);
                    while (mis) {
                        const int p = p0 + __ffs((int)mis) - 1; mis &= mis - 1u;
                        const uint32_t r_ = (uint32_t)(p - last - 1);
                        if (first < 0) { first = p; md_rest += 1; } else md_rest += d_ndig(r_) + 1;
                        cs_b += (r_ ? 1 + d_ndig(r_) : 0) + 3;
                        last = p;
                    }
                }
                if (first >= 0) { ev = true; lead = (uint32_t)first; trail = (uint32_t)(L - 1 - last); } else tot = (uint32_t)L;
                const uint32_t tail = (uint32_t)(L - 1 - last);
                if (tail) cs_b += 1 + d_ndig(tail);
            } else if (op == 1) { nm += L; nI += L; cs_b = 1 + L; }
            else { nm += L; nD += L; ev = true; md_rest = 1 + L; cs_b = 1 + L; }
        }
        // ---- matching columns since the last MD event before this lane
        uint32_t s = ev ? trail : tot; int f = ev ? 1 : 0;
        if (lane == 0 && !ev) s += carry;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t ps = (uint32_t)__shfl_up((int)s, o); const int pf = __shfl_up(f, o);
            if (lane >= o) { if (!f) s += ps; f |= pf; }
        }
        uint32_t run_in = (uint32_t)__shfl_up((int)s, 1);
        if (lane == 0) run_in = carry;
        carry = (uint32_t)__shfl((int)s, 63);
        int md_b = (ev && want_md) ? d_ndig(run_in + lead) + md_rest : 0;
        if (!want_cs) cs_b = 0;
        if (!WRITE) { md_sum += md_b; cs_sum += cs_b; }
        else {
            const int mdi = d_wave_incl(md_b, lane), csi = d_wave_incl(cs_b, lane);
            ByteW mp, cp; mp.init(md_p + md_base + (uint32_t)(mdi - md_b)); cp.init(cs_p + cs_base + (uint32_t)(csi - cs_b));
            md_base += (uint32_t)__shfl(mdi, 63); cs_base += (uint32_t)__shfl(csi, 63);
            if (have) {
                if (op == 0) {
                    int last = -1; bool firstev = true;
                    for (int p0 = 0; p0 < L; p0 += 32) {
                        const B32 qw = p0 == 0 ? qw0 : d_strand32(A.q2, A.qn, qb0, ql, rev, qi + p0), tw = p0 == 0 ? tw0 : d_fetch32(A.t2, A.tn, tb0 + ti + p0);
                        uint32_t mis = d_mis32(qw, tw, L - p0);
                        while (mis) {
                            const int j = __ffs((int)mis) - 1, p = p0 + j; mis &= mis - 1u;
                            const uint32_t r_ = (uint32_t)(p - last - 1);
                            const uint8_t tc = d_base_char(tw, j), qc = d_base_char(qw, j);
                            if (want_md) { mp.put_uint(firstev ? run_in + r_ : r_); mp.put(tc); }
                            if (want_cs) { if (r_) { cp.put(':'); cp.put_uint(r_); } cp.put('*'); cp.put(tc | 32u); cp.put(qc | 32u); }
                            firstev = false; last = p;
                        }
                    }
                    const uint32_t tail = (uint32_t)(L - 1 - last);
                    if (want_cs && tail) { cp.put(':'); cp.put_uint(tail); }
                } else if (op == 1) {
                    if (want_cs) {
                        cp.put('+');
                        for (int p0 = 0; p0 < L; p0 += 32) { const B32 qw = d_strand32(A.q2, A.qn, qb0, ql, rev, qi + p0); const int n = L - p0 < 32 ? L - p0 : 32; for (int j = 0; j < n; ++j) cp.put(d_base_char(qw, j) | 32u); }
                    }
                } else {
                    if (want_md) { mp.put_uint(run_in); mp.put('^'); }
                    if (want_cs) cp.put('-');
                    for (int p0 = 0; p0 < L; p0 += 32) {
                        const B32 tw = d_fetch32(A.t2, A.tn, tb0 + ti + p0); const int n = L - p0 < 32 ? L - p0 : 32;
                        for (int j = 0; j < n; ++j) { const uint8_t tc = d_base_char(tw, j); if (want_md) mp.put(tc); if (want_cs) cp.put(tc | 32u); }
                    }
                }
                mp.flush(); cp.flush();
            }
        }
        qi0 += __shfl(qinc, 63); ti0 += __shfl(tinc, 63);
    }
    if (!WRITE) {
        out.nm = d_wave_sum(nm); out.nI = d_wave_sum(nI); out.nD = d_wave_sum(nD);
        out.md_len = want_md ? d_wave_sum(md_sum) + d_ndig(carry) : 0;
        out.cs_len = d_wave_sum(cs_sum);
    } else if (want_md && lane == 0) d_put_uint(md_p + md_base, carry);
}

__global__ void __launch_bounds__(64) k_bam_scan(BamArgs A)
{
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= A.n_mapped) return;
    const telr_aln a = A.alns[k];
    BamInfo I;
    d_bam_walk<false>(A, a, lane, nullptr, nullptr, I);
    if (lane == 0) A.info[k] = I;
}

// the SA tag of record k: the other primary / supplementary records of the read, `rname,pos,strand,CIGAR,mapQ,NM;` each
// (CIGAR reduced to clip / M / I / D totals as minimap2 prints it).  p == nullptr: length only.
__device__ __forceinline__ uint8_t *d_put_str(uint8_t *p, const char *s, int n) { for (int i = 0; i < n; ++i) p[i] = (uint8_t)s[i]; return p + n; }
__device__ int d_bam_sa(const BamArgs &A, int k, uint8_t *p)
{
    const telr_aln &a = A.alns[k];
    int i0 = k, i1 = k + 1;
    while (i0 > 0 && A.alns[i0 - 1].qid == a.qid) --i0;
    while (i1 < A.n_mapped && A.alns[i1].qid == a.qid) ++i1;
    int n = 0;
    for (int k2 = i0; k2 < i1; ++k2) {
        const telr_aln &b = A.alns[k2];
        if (k2 == k || (b.flags & TELR_F_SECONDARY)) continue;
        const bool brev = (b.flags & TELR_F_REV) != 0;
        const int ql = b.qlen, b5 = brev ? ql - b.qe : b.qs, b3 = brev ? b.qs : ql - b.qe;
        const int nI = A.info[k2].nI, nD = A.info[k2].nD, M = (b.qe - b.qs) - nI, nmm = b.blen - b.mlen;
        const int tn0 = A.tname_off[b.tid], tnl = A.tname_off[b.tid + 1] - tn0 - 1;
        n += tnl + 1 + d_ndig((uint32_t)(b.ts + 1)) + 3 + (b5 ? d_ndig((uint32_t)b5) + 1 : 0) + d_ndig((uint32_t)M) + 1 + (nI ? d_ndig((uint32_t)nI) + 1 : 0)
             + (nD ? d_ndig((uint32_t)nD) + 1 : 0) + (b3 ? d_ndig((uint32_t)b3) + 1 : 0) + 1 + d_ndig((uint32_t)b.mapq) + 1 + d_ndig((uint32_t)nmm) + 1;
        if (p) {
            p = d_put_str(p, A.tnames + tn0, tnl); *p++ = ',';
            p = d_put_uint(p, (uint32_t)(b.ts + 1)); *p++ = ','; *p++ = brev ? '-' : '+'; *p++ = ',';
            if (b5) { p = d_put_uint(p, (uint32_t)b5); *p++ = 'S'; }
            p = d_put_uint(p, (uint32_t)M); *p++ = 'M';
            if (nI) { p = d_put_uint(p, (uint32_t)nI); *p++ = 'I'; }
            if (nD) { p = d_put_uint(p, (uint32_t)nD); *p++ = 'D'; }
            if (b3) { p = d_put_uint(p, (uint32_t)b3); *p++ = 'S'; }
            *p++ = ','; p = d_put_uint(p, (uint32_t)b.mapq); *p++ = ','; p = d_put_uint(p, (uint32_t)nmm); *p++ = ';';
        }
    }
    return n;
}

// field sizes of a record (shared by the size and the write kernel)
struct BamLayout { int l_name, clip5, clip3, hard, n_cig, long_cigar, seq_lo, l_seq, sec, sup, rev; };
__device__ __forceinline__ BamLayout d_bam_layout(const BamArgs &A, const telr_aln &a)
{
    BamLayout Y;
    Y.l_name = (int)(A.qname_off[a.qid + 1] - A.qname_off[a.qid]);
    if (a.tid < 0) { Y.clip5 = Y.clip3 = Y.hard = Y.n_cig = Y.long_cigar = Y.seq_lo = Y.sec = Y.sup = Y.rev = 0; Y.l_seq = a.qlen; return Y; }
    Y.rev = (a.flags & TELR_F_REV) ? 1 : 0; Y.sec = (a.flags & TELR_F_SECONDARY) ? 1 : 0; Y.sup = (a.flags & TELR_F_SUPPL) ? 1 : 0;
    Y.clip5 = Y.rev ? a.qlen - a.qe : a.qs; Y.clip3 = Y.rev ? a.qs : a.qlen - a.qe;
    Y.hard = Y.sup && !(A.flags & TELR_SAM_SOFTCLIP);
    Y.n_cig = a.n_cigar + (Y.clip5 > 0) + (Y.clip3 > 0);
    Y.long_cigar = Y.n_cig > 65535;
    Y.seq_lo = Y.sec ? 0 : (Y.hard ? Y.clip5 : 0);
    Y.l_seq = Y.sec ? 0 : (Y.hard ? a.qlen - Y.clip5 - Y.clip3 : a.qlen);
    return Y;
}

__global__ void __launch_bounds__(256) k_bam_size(BamArgs A)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= A.nrec) return;
    const telr_aln a = A.alns[k];
    const BamLayout Y = d_bam_layout(A, a);
    const int rg = A.rg_len ? 3 + A.rg_len + 1 : 0;
    uint32_t size;
    if (A.emit && !A.emit[k]) {          // another rank writes this record: no bytes here, sorted behind everything
        A.rec_size[k] = 0; A.key[k] = ~0ULL;
        return;
    }
    if (a.tid < 0) {
        size = 36u + Y.l_name + (uint32_t)(Y.l_seq + 1) / 2 + Y.l_seq + rg;
        A.key[k] = ~0ULL - 1;
    } else {
        const BamInfo I = A.info[k];
        const int sa = Y.sec ? 0 : d_bam_sa(A, k, nullptr);
        size = 36u + Y.l_name + 4u * (Y.long_cigar ? 2 : Y.n_cig) + (uint32_t)(Y.l_seq + 1) / 2 + Y.l_seq + 14
               + ((A.flags & TELR_SAM_MD) ? 3 + I.md_len + 1 : 0) + ((A.flags & TELR_SAM_CS) ? 3 + I.cs_len + 1 : 0) + (sa ? 3 + sa + 1 : 0)
               + 4 + 7 + 7 + (Y.sec ? 0 : 7) + rg + (Y.long_cigar ? 8 + 4 * Y.n_cig : 0);
        A.key[k] = (uint64_t)(a.tid + 1) << 33 | (uint64_t)(uint32_t)a.ts << 1 | (uint64_t)Y.rev;     // samtools sort: refID, pos, then forward before reverse
    }
    A.rec_size[k] = size;
}

__global__ void __launch_bounds__(256) k_bam_gather_sizes(const uint32_t *__restrict__ rec_size, const uint32_t *__restrict__ order, int32_t n, uint64_t *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = rec_size[order[i]]; else if (i == n) out[i] = 0;
}
__global__ void __launch_bounds__(256) k_bam_scatter_off(const uint64_t *__restrict__ ustart_sorted, const uint32_t *__restrict__ order, int32_t n, uint64_t head, uint64_t *__restrict__ rec_ustart)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) rec_ustart[order[i]] = ustart_sorted[i] + head;
}
__global__ void __launch_bounds__(256) k_iota_u32(uint32_t *__restrict__ v, int32_t n) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) v[i] = (uint32_t)i; }

__device__ __forceinline__ uint8_t *d_tag_i(uint8_t *p, char a, char b, int32_t v) { p[0] = (uint8_t)a; p[1] = (uint8_t)b; p[2] = 'i'; d_st32(p + 3, (uint32_t)v); return p + 7; }

__global__ void __launch_bounds__(64) k_bam_write(BamArgs A)
{
    const int lane = threadIdx.x;
    if ((int)blockIdx.x + A.s0 >= A.nrec) return;
    const int k = A.order ? (int)A.order[blockIdx.x + A.s0] : (int)blockIdx.x;
    const telr_aln a = A.alns[k];
    const BamLayout Y = d_bam_layout(A, a);
    const uint32_t size = A.rec_size[k];
    if (size == 0) return;               // slice mode: not this rank's record
    uint8_t *const rec = A.ubuf + A.rec_ustart[k];
    const bool un = a.tid < 0;
    const int n_cig_field = un ? 0 : (Y.long_cigar ? 2 : Y.n_cig);
    uint8_t *const p_name = rec + 36, *const p_cig = p_name + Y.l_name, *const p_seq = p_cig + 4 * n_cig_field,
            *const p_qual = p_seq + (Y.l_seq + 1) / 2, *const p_tags = p_qual + Y.l_seq;
    BamInfo I = {0, 0, 0, 0, 0};
    if (!un) I = A.info[k];
    const bool want_md = (A.flags & TELR_SAM_MD) != 0, want_cs = (A.flags & TELR_SAM_CS) != 0;
    // tag offsets (mapped records)
    uint8_t *p_md = p_tags + 14, *p_cs = p_md + (want_md ? 3 + I.md_len + 1 : 0), *p_sa = p_cs + (want_cs ? 3 + I.cs_len + 1 : 0);
    if (lane == 0) {
        d_st32(rec, size - 4u);
        d_st32(rec + 4, un ? 0xffffffffu : (uint32_t)a.tid); d_st32(rec + 8, un ? 0xffffffffu : (uint32_t)a.ts);
        rec[12] = (uint8_t)Y.l_name; rec[13] = un ? 0 : (uint8_t)a.mapq;
        d_st16(rec + 14, un ? 4680u : (uint32_t)d_reg2bin(a.ts, a.te > a.ts ? a.te : a.ts + 1));
        d_st16(rec + 16, (uint32_t)n_cig_field);
        d_st16(rec + 18, un ? 4u : (uint32_t)((Y.rev ? 0x10 : 0) | (Y.sec ? 0x100 : 0) | (Y.sup ? 0x800 : 0)));
        d_st32(rec + 20, (uint32_t)Y.l_seq); d_st32(rec + 24, 0xffffffffu); d_st32(rec + 28, 0xffffffffu); d_st32(rec + 32, 0u);
    }
    { const char *nm = A.qnames + A.qname_off[a.qid]; for (int i = lane; i < Y.l_name; i += 64) p_name[i] = (uint8_t)nm[i]; }
    // CIGAR (with clips), or the -L placeholder plus the real one in CG:B,I at the end of the tags
    if (!un) {
        const uint32_t clipop = Y.hard ? 5u : 4u;
        uint8_t *dst = Y.long_cigar ? rec + size - 4 * Y.n_cig : p_cig;
        const uint32_t *__restrict__ cg = A.cig + a.cigar_off;
        const int lead = Y.clip5 > 0 ? 1 : 0;
        for (int i = lane; i < Y.n_cig; i += 64) {
            uint32_t v;
            if (i < lead) v = (uint32_t)Y.clip5 << 4 | clipop; else if (i - lead < a.n_cigar) v = cg[i - lead]; else v = (uint32_t)Y.clip3 << 4 | clipop;
            d_st32(dst + 4 * i, v);
        }
        if (Y.long_cigar && lane == 0) {
            d_st32(p_cig, (uint32_t)Y.l_seq << 4 | 4u); d_st32(p_cig + 4, (uint32_t)(a.te - a.ts) << 4 | 3u);
            uint8_t *t = rec + size - 4 * Y.n_cig - 8; t[0] = 'C'; t[1] = 'G'; t[2] = 'B'; t[3] = 'I'; d_st32(t + 4, (uint32_t)Y.n_cig);
        }
    }
    // SEQ: 32 bases (16 bytes) per lane and trip; QUAL: 0xff
    {
        const int64_t qb0 = A.qboff[a.qid];
        for (int i = lane * 32; i < Y.l_seq; i += 64 * 32) {
            const B32 w = d_strand32(A.q2, A.qn, qb0, a.qlen, Y.rev, Y.seq_lo + i);
            const int n = Y.l_seq - i < 32 ? Y.l_seq - i : 32;
            uint8_t *d = p_seq + (i >> 1);
            if (n == 32) {
#pragma unroll
                for (int g = 0; g < 4; ++g) d_st32(d + 4 * g, d_nib8((uint32_t)(w.w >> (16 * g)), w.n >> (8 * g)));
            } else {
                for (int g = 0; g * 8 < n; ++g) {
                    const int m = n - g * 8 < 8 ? n - g * 8 : 8;
                    uint32_t c = (uint32_t)(w.w >> (16 * g)) & 0xffffu, nn = (w.n >> (8 * g)) & 0xffu;
                    if (m < 8) { c &= (1u << (2 * m)) - 1u; nn &= (1u << m) - 1u; }
                    uint32_t v = d_nib8(c, nn);
                    // bases beyond m come out as 'A' (code 1): clear them
                    for (int b = 0; b < (m + 1) / 2; ++b) { uint32_t byte = (v >> (8 * b)) & 0xffu; if (2 * b + 1 >= m) byte &= 0xf0u; d[4 * g + b] = (uint8_t)byte; }
                }
            }
        }
        const int head = (int)((4 - ((uintptr_t)p_qual & 3)) & 3), nh = head < Y.l_seq ? head : Y.l_seq;
        if (lane < nh) p_qual[lane] = 0xff;
        const int body = (Y.l_seq - nh) >> 2;
        uint32_t *q4 = (uint32_t*)(p_qual + nh);
        for (int i = lane; i < body; i += 64) q4[i] = 0xffffffffu;
        const int tail0 = nh + body * 4;
        if (tail0 + lane < Y.l_seq && lane < 4) p_qual[tail0 + lane] = 0xff;
    }
    if (un) {
        if (A.rg_len && lane == 0) { uint8_t *t = p_tags; t[0] = 'R'; t[1] = 'G'; t[2] = 'Z'; t = d_put_str(t + 3, A.rg, A.rg_len); *t = 0; }
        return;
    }
    // tags: NM AS [MD] [cs] [SA] tp cm s1 [s2] [RG] [CG]
    BamInfo dummy;
    d_bam_walk<true>(A, a, lane, p_md + 3, p_cs + 3, dummy);
    if (lane == 0) {
        uint8_t *t = d_tag_i(p_tags, 'N', 'M', I.nm); d_tag_i(t, 'A', 'S', a.dp_score);
        if (want_md) { p_md[0] = 'M'; p_md[1] = 'D'; p_md[2] = 'Z'; p_md[3 + I.md_len] = 0; }
        if (want_cs) { p_cs[0] = 'c'; p_cs[1] = 's'; p_cs[2] = 'Z'; p_cs[3 + I.cs_len] = 0; }
        t = p_sa;
        if (!Y.sec) {
            const int sa = d_bam_sa(A, k, nullptr);
            if (sa) { t[0] = 'S'; t[1] = 'A'; t[2] = 'Z'; d_bam_sa(A, k, t + 3); t[3 + sa] = 0; t += 3 + sa + 1; }
        }
        t[0] = 't'; t[1] = 'p'; t[2] = 'A'; t[3] = Y.sec ? 'S' : 'P'; t += 4;
        t = d_tag_i(t, 'c', 'm', a.cnt); t = d_tag_i(t, 's', '1', a.score);
        if (!Y.sec) t = d_tag_i(t, 's', '2', a.subsc);
        if (A.rg_len) { t[0] = 'R'; t[1] = 'G'; t[2] = 'Z'; t = d_put_str(t + 3, A.rg, A.rg_len); *t++ = 0; }
    }
}

// ---- CRC-32 (IEEE, reflected 0xEDB88320) of a BGZF block by 256 threads: every thread its own piece, then
// crc(A || B) = crc(A) * x^(8|B|) + crc(B) over GF(2) (the identity behind zlib's crc32_combine)
struct CrcTabs { uint32_t byte_tab[256]; uint32_t xpow255[256]; uint32_t xpow64[1024]; };      // xpow255[k] = x^(8 * 255 * k) mod P, xpow64[k] = x^(8 * 64 * k)
__device__ __forceinline__ uint32_t d_gf2_mulmod(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m; m >>= 1) { if (a & m) p ^= b; b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1; }
    return p;
}
// block-wide: CRC-32 of n bytes in LDS (n == BAM_BLK: pieces of 255 bytes; otherwise thread 0 alone).  All 256 threads call; result valid in thread 0.
__device__ __forceinline__ uint32_t d_block_crc(const uint8_t *sh, int n, const uint32_t *tab, const uint32_t *xp, uint32_t *red)
{
    const int t = threadIdx.x;
    uint32_t c = 0;
    if (n == BAM_BLK) {
        uint32_t s = 0xffffffffu;
        const uint8_t *p = sh + t * 255;
        for (int i = 0; i < 255; ++i) s = tab[(s ^ p[i]) & 0xffu] ^ (s >> 8);
        c = d_gf2_mulmod(xp[255 - t], ~s);
    } else if (t == 0) {
        uint32_t s = 0xffffffffu;
        for (int i = 0; i < n; ++i) s = tab[(s ^ sh[i]) & 0xffu] ^ (s >> 8);
        c = ~s;
    }
    for (int s = 32; s >= 1; s >>= 1) c ^= (uint32_t)__shfl_xor((int)c, s);
    if ((t & 63) == 0) red[t >> 6] = c;
    __syncthreads();
    return red[0] ^ red[1] ^ red[2] ^ red[3];
}

// level 0: one stored deflate block per BGZF block.  out block b at b * (BAM_BLK + 31).
__global__ void __launch_bounds__(256) k_bgzf_store(const uint8_t *__restrict__ ubuf, uint64_t utotal, const CrcTabs *__restrict__ T, uint8_t *__restrict__ cbuf)
{
    __shared__ uint32_t sh4[BAM_BLK / 4];
    __shared__ uint32_t tab[256], xp[256], red[4];
    const int t = threadIdx.x;
    const uint64_t b = blockIdx.x, u0 = b * BAM_BLK;
    const int n = (int)(utotal - u0 < BAM_BLK ? utotal - u0 : BAM_BLK);
    tab[t] = T->byte_tab[t]; xp[t] = T->xpow255[t];
    uint8_t *sh = (uint8_t*)sh4;
    uint8_t *out = cbuf + b * (uint64_t)(BAM_BLK + 31);
    // the stream is read in aligned dwords (its start u0 is a multiple of 4: BAM_BLK is)
    const uint32_t *src4 = (const uint32_t*)(ubuf + u0);
    const int n4 = (n + 3) >> 2;
    for (int i = t; i < n4; i += 256) sh4[i] = src4[i];          // ubuf is padded to a multiple of 4 by the host
    __syncthreads();
    // payload at out + 23 (unaligned by one byte: byte-granular dword stores)
    for (int i = t; i < (n >> 2); i += 256) d_st32(out + 23 + 4 * i, sh4[i]);
    for (int i = (n & ~3) + t; i < n; i += 256) out[23 + i] = sh[i];
    const uint32_t crc = d_block_crc(sh, n, tab, xp, red);
    if (t == 0) {
        const uint8_t hdr[16] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0 };
        for (int i = 0; i < 16; ++i) out[i] = hdr[i];
        d_st16(out + 16, (uint32_t)(n + 31 - 1));
        out[18] = 1; d_st16(out + 19, (uint32_t)n); d_st16(out + 21, (uint32_t)(~n & 0xffff));
        d_st32(out + 23 + n, crc); d_st32(out + 27 + n, (uint32_t)n);
    }
}

// ---- level >= 1: deflate on the device --------------------------------------------------------------------------------
// A BAM stream is four kinds of bytes with very different statistics: binary fixed fields + CIGAR words (class A), the
// 4-bit SEQ whose bytes take 16 values followed by the QUAL run of 0xff (class B), and tag text -- MD / cs / SA -- (class C).
// One Huffman table over the mixture pays ~1 bit per byte for not knowing the field; so every field segment becomes its own
// DEFLATE block (RFC 1951, BTYPE = 10) coded with the table of its class.  The three tables are computed ONCE per file
// (k_bam_hist over a sample of the blocks, Huffman lengths on the host) and their headers are pasted as ready-made bit strings.
// Matching is run-length only (distance 1): QUAL collapses, CIGAR zero bytes shorten; SEQ and the text have no long repeats
// worth a hash table.  Field boundaries come from the records themselves (l_read_name, n_cigar_op, l_seq in the stream).
#define DEFL_NCLS    3
#define DEFL_THREADS 1024
#define DEFL_PIECE   64             /* bytes per thread: BAM_BLK = 1020 pieces */
#define DEFL_SEGCAP  256
#define DEFL_MINSEG  192            /* a field shorter than this is coded with the table in force (a switch costs 40-80 bytes) */
#define DEFL_SLOT    65536
struct DeflTabs {
    uint32_t lit[DEFL_NCLS][288];       // symbols 0..285: bit-reversed code | length << 16
    uint32_t hdr_bits[DEFL_NCLS];       // bits of the dynamic-block header (after BFINAL / BTYPE)
    uint32_t hdr[DEFL_NCLS][96];
    uint16_t len_sym[259];              // match length -> (symbol - 257) | extra bits << 5 | extra value << 8
};
struct DeflSeg { uint16_t start; uint16_t cls; };

// segments of block [u0, u0 + n): where the table changes.  Executed by wave 0 of the workgroup; result in LDS (seg[0].start = 0).
__device__ __forceinline__ int d_defl_segments(const uint8_t *__restrict__ ubuf, uint64_t u0, int n, uint64_t head, const uint64_t *__restrict__ ust, int32_t nrec,
                                               int32_t rec0, DeflSeg *seg, int lane)
{
    // class at u0
    int nseg = 0;
    int r = rec0;                                     // last record starting at or before u0 (-1: inside the header)
    if (r < 0) { if (lane == 0) { seg[0].start = 0; seg[0].cls = 2; } nseg = 1; r = 0; if (nrec == 0) return 1; }
    const uint64_t uend = u0 + (uint64_t)n;
    bool first = rec0 >= 0;
    for (; r < nrec; r += 64) {
        const int k = r + lane;
        uint64_t s = ~0ULL, p1 = 0, p2 = 0, e = 0;
        if (k < nrec) s = ust[k] + head;
        const bool in = k < nrec && s < uend;
        if (in) {
            e = ust[k + 1] + head;
            const uint8_t *h = ubuf + s;
            const uint32_t l_name = h[12], n_cig = (uint32_t)h[16] | (uint32_t)h[17] << 8, l_seq = (uint32_t)h[20] | (uint32_t)h[21] << 8 | (uint32_t)h[22] << 16 | (uint32_t)h[23] << 24;
            p1 = s + 36 + l_name + 4ull * n_cig; p2 = p1 + (l_seq + 1) / 2 + l_seq;
        }
        if (first) {      // lane 0 holds the record around u0
            if (lane == 0) { seg[0].start = 0; seg[0].cls = (uint16_t)(u0 < p1 ? 0 : (u0 < p2 ? 1 : 2)); }
            nseg = 1; first = false;
        }
        // candidate switches: (s, A), (p1, B), (p2, C) strictly inside the block, whose own field is long enough
        bool k0 = in && s > u0 && s < uend && (p1 - s) >= DEFL_MINSEG, k1 = in && p1 > u0 && p1 < uend && (p2 - p1) >= DEFL_MINSEG, k2 = in && p2 > u0 && p2 < uend && (e - p2) >= DEFL_MINSEG;
        const int cnt = (int)k0 + (int)k1 + (int)k2;
        const int inc = d_wave_incl(cnt, lane);
        int w = nseg + inc - cnt;
        if (k0 && w < DEFL_SEGCAP) { seg[w].start = (uint16_t)(s - u0); seg[w].cls = 0; ++w; }
        if (k1 && w < DEFL_SEGCAP) { seg[w].start = (uint16_t)(p1 - u0); seg[w].cls = 1; ++w; }
        if (k2 && w < DEFL_SEGCAP) { seg[w].start = (uint16_t)(p2 - u0); seg[w].cls = 2; ++w; }
        nseg += __shfl(inc, 63);
        if (nseg > DEFL_SEGCAP) nseg = DEFL_SEGCAP;
        if (__ballot(k < nrec && s >= uend) || r + 64 >= nrec) break;
    }
    return nseg;
}

// first record of every block: the last one starting at or before the block's first byte (-1: none, the block starts in the header)
__global__ void __launch_bounds__(256) k_blk_first_rec(const uint64_t *__restrict__ ust, int32_t nrec, uint64_t head, int32_t nblk, int32_t *__restrict__ rec0)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nblk) return;
    const uint64_t u0 = (uint64_t)b * BAM_BLK;
    if (u0 < head || nrec == 0) { rec0[b] = -1; return; }
    int lo = 0, hi = nrec - 1;            // ust[0] + head = head <= u0
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ust[mid] + head <= u0) lo = mid; else hi = mid - 1; }
    rec0[b] = lo;
}

// A thread walks its own 64-byte piece of the block: with the pieces end to end in LDS the 64 lanes of a wave read addresses 64
// bytes apart -- four of the 64 banks, a 16-way conflict on every byte.  The staged block therefore carries one pad dword per
// piece (stride 17 dwords: conflict-free); byte i of the block lives at i + 4 * (i / 64).
#define DEFL_INB(in_, i_) ((in_)[(i_) + (((i_) >> 6) << 2)])
#define DEFL_IN4_WORDS (BAM_BLK / 4 + BAM_BLK / 64 + 4)
// The token walk of one thread over its piece.  MODE 0: bits; 1: emit into the LDS bit buffer; 2: histogram (hist[cls][sym]).
struct BitW { uint64_t acc; int nb; uint32_t word; uint32_t *out; };
__device__ __forceinline__ void d_bw_put(BitW &W, uint32_t v, int len)
{
    W.acc |= (uint64_t)v << W.nb; W.nb += len;
    if (W.nb >= 32) { atomicOr(&W.out[W.word], (uint32_t)W.acc); ++W.word; W.acc >>= 32; W.nb -= 32; }
}
template <int MODE>
__device__ __forceinline__ uint32_t d_defl_piece(const uint8_t *in, int n, int t, const DeflSeg *seg, int nseg, const DeflTabs *T, BitW *W, uint32_t *hist,
                                                 const uint32_t *lit /* [DEFL_NCLS][288], a copy of T->lit in LDS (unused by MODE 2) */, const uint16_t *len_sym /* [259], in LDS */)
{
    const int p0 = t * DEFL_PIECE;
    if (p0 >= n) return 0;
    const int p1 = p0 + DEFL_PIECE < n ? p0 + DEFL_PIECE : n;
    // segment in force just before p0 (a switch exactly at p0 is this thread's to emit)
    int lo = 0, hi = nseg - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)seg[mid].start < p0) lo = mid; else hi = mid - 1; }
    int si = lo, cls = seg[si].cls;
    int next = si + 1 < nseg ? (int)seg[si + 1].start : 0x7fffffff;
    uint32_t bits = 0;
    auto open_block = [&](int s) {
        const int c = seg[s].cls;
        if (MODE == 0) bits += 3 + T->hdr_bits[c];
        else if (MODE == 1) {
            d_bw_put(*W, (s == nseg - 1 ? 1u : 0u) | 2u << 1, 3);
            const uint32_t hb = T->hdr_bits[c];
            for (uint32_t i = 0; i < hb; i += 16) { const uint32_t w = T->hdr[c][i >> 5] >> (i & 31); const int l = hb - i < 16 ? (int)(hb - i) : 16; d_bw_put(*W, w & ((1u << l) - 1u), l); }
        }
    };
    auto put_sym = [&](int c, int sym) {
        if (MODE == 0) bits += lit[c * 288 + sym] >> 16;
        else if (MODE == 1) { const uint32_t e = lit[c * 288 + sym]; d_bw_put(*W, e & 0xffffu, (int)(e >> 16)); }
        else atomicAdd(&hist[c * 288 + sym], 1u);
    };
    if (p0 == 0) open_block(0);
    int i = p0;
    uint32_t prev = i > 0 ? DEFL_INB(in, i - 1) : 0x100u;
    while (i < p1) {
        if (i == next) {       // the table changes here
            put_sym(cls, 256);
            ++si; cls = seg[si].cls; open_block(si);
            next = si + 1 < nseg ? (int)seg[si + 1].start : 0x7fffffff;
        }
        const uint32_t b = DEFL_INB(in, i);
        const int lim = p1 < next ? p1 : next;
        if (b == prev) {
            int L = 1;
            while (i + L < lim && DEFL_INB(in, i + L) == b) ++L;
            if (L >= 3) {
                if (MODE == 2) { atomicAdd(&hist[cls * 288 + 257 + (len_sym[L] & 31)], 1u); }
                else {
                    const uint32_t ls = len_sym[L]; const int sym = 257 + (int)(ls & 31), eb = (int)(ls >> 5 & 7);
                    put_sym(cls, sym);
                    if (MODE == 0) bits += eb + 1; else { if (eb) d_bw_put(*W, ls >> 8, eb); d_bw_put(*W, 0u, 1); }      // the only distance code (distance 1) is one bit
                }
                i += L; continue;            // prev stays b
            }
        }
        put_sym(cls, (int)b);
        prev = b; ++i;
    }
    if (p1 == n) put_sym(cls, 256);
    return bits;
}

__global__ void __launch_bounds__(DEFL_THREADS) k_bam_hist(const uint8_t *__restrict__ ubuf, uint64_t utotal, uint64_t head, const uint64_t *__restrict__ ust, int32_t nrec,
                                                          const int32_t *__restrict__ rec0, int32_t stride, const DeflTabs *__restrict__ T, uint32_t *__restrict__ ghist)
{
    __shared__ uint32_t in4[DEFL_IN4_WORDS];
    __shared__ DeflSeg seg[DEFL_SEGCAP];
    __shared__ uint32_t hist[DEFL_NCLS * 288];
    __shared__ uint16_t s_len[260];
    __shared__ int s_nseg;
    const int t = threadIdx.x;
    if (t < 259) s_len[t] = T->len_sym[t];
    const uint64_t b = (uint64_t)blockIdx.x * stride, u0 = b * BAM_BLK;
    const int n = (int)(utotal - u0 < BAM_BLK ? utotal - u0 : BAM_BLK);
    const uint32_t *src4 = (const uint32_t*)(ubuf + u0);
    for (int i = t; i < (n + 3) >> 2; i += DEFL_THREADS) in4[i + (i >> 4)] = src4[i];
    for (int i = t; i < DEFL_NCLS * 288; i += DEFL_THREADS) hist[i] = 0;
    if (t < 64) { const int ns = d_defl_segments(ubuf, u0, n, head, ust, nrec, rec0[b], seg, t); if (t == 0) s_nseg = ns; }
    __syncthreads();
    d_defl_piece<2>((const uint8_t*)in4, n, t, seg, s_nseg, T, nullptr, hist, nullptr, s_len);
    __syncthreads();
    for (int i = t; i < DEFL_NCLS * 288; i += DEFL_THREADS) if (hist[i]) atomicAdd(&ghist[i], hist[i]);
}

// one BGZF block: out slot b (DEFL_SLOT bytes), its size in csize[b]
__global__ void __launch_bounds__(DEFL_THREADS) k_bgzf_deflate(const uint8_t *__restrict__ ubuf, uint64_t utotal, uint64_t head, const uint64_t *__restrict__ ust, int32_t nrec,
                                                              const int32_t *__restrict__ rec0, const DeflTabs *__restrict__ T, const CrcTabs *__restrict__ CT,
                                                              uint8_t *__restrict__ slots, uint32_t *__restrict__ csize, uint32_t blk0)
{
    __shared__ uint32_t in4[DEFL_IN4_WORDS];
    __shared__ uint32_t out4[BAM_BLK / 4 + 8];
    __shared__ DeflSeg seg[DEFL_SEGCAP];
    __shared__ uint32_t tab[256], wsum[16], red[4];
    __shared__ uint32_t s_lit[DEFL_NCLS * 288];          // the code tables next to the data: a lookup per input byte, twice
    __shared__ uint16_t s_len[260];
    __shared__ int s_nseg; __shared__ uint32_t s_total;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int i = t; i < DEFL_NCLS * 288; i += DEFL_THREADS) s_lit[i] = T->lit[i / 288][i % 288];
    if (t < 259) s_len[t] = T->len_sym[t];
    const uint64_t b = (uint64_t)blockIdx.x + blk0, u0 = b * BAM_BLK;
    const int n = (int)(utotal - u0 < BAM_BLK ? utotal - u0 : BAM_BLK);
    const uint32_t *src4 = (const uint32_t*)(ubuf + u0);
    for (int i = t; i < (n + 3) >> 2; i += DEFL_THREADS) in4[i + (i >> 4)] = src4[i];
    for (int i = t; i < BAM_BLK / 4 + 8; i += DEFL_THREADS) out4[i] = 0;
    if (t < 256) tab[t] = CT->byte_tab[t];
    if (t < 64) { const int ns = d_defl_segments(ubuf, u0, n, head, ust, nrec, rec0[b], seg, t); if (t == 0) s_nseg = ns; }
    __syncthreads();
    const uint8_t *in = (const uint8_t*)in4;
    const int nseg = s_nseg;
    const uint32_t bits = d_defl_piece<0>(in, n, t, seg, nseg, T, nullptr, nullptr, s_lit, s_len);
    // exclusive scan of the bit counts over the workgroup
    const uint32_t inc = (uint32_t)d_wave_incl((int)bits, lane);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    if (t < 16) { uint32_t v = wsum[t]; for (int o = 1; o < 16; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)v, o, 16); if (t >= o) v += y; } wsum[t] = v; if (t == 15) s_total = v; }
    __syncthreads();
    const uint32_t off = inc - bits + (wv ? wsum[wv - 1] : 0u), total = s_total;
    const uint32_t cbytes = (total + 7) >> 3;
    const bool stored = cbytes + 26 > (uint32_t)n + 31 || cbytes > BAM_BLK;        // deflate no smaller than a stored block: keep it stored
    if (!stored) {
        BitW W; W.acc = 0; W.nb = (int)(off & 31); W.word = off >> 5; W.out = out4;
        d_defl_piece<1>(in, n, t, seg, nseg, T, &W, nullptr, s_lit, s_len);
        if (W.nb) atomicOr(&out4[W.word], (uint32_t)W.acc);
    }
    // CRC-32 of the input: pieces of 64 bytes, combined with x^(512 k) (crc(A || B) = crc(A) x^(8|B|) + crc(B))
    uint32_t c = 0;
    if (n == BAM_BLK) {
        if (t < BAM_BLK / DEFL_PIECE) {
            uint32_t s = 0xffffffffu;
            const uint8_t *p = in + t * (DEFL_PIECE + 4);            // the padded piece
            for (int i = 0; i < DEFL_PIECE; ++i) s = tab[(s ^ p[i]) & 0xffu] ^ (s >> 8);
            c = d_gf2_mulmod(CT->xpow64[BAM_BLK / DEFL_PIECE - 1 - t], ~s);
        }
    } else if (t == 0) {
        uint32_t s = 0xffffffffu;
        for (int i = 0; i < n; ++i) s = tab[(s ^ DEFL_INB(in, i)) & 0xffu] ^ (s >> 8);
        c = ~s;
    }
    for (int s = 32; s >= 1; s >>= 1) c ^= (uint32_t)__shfl_xor((int)c, s);
    __syncthreads();                     // out4 complete, wsum free
    if (lane == 0) wsum[wv] = c;
    __syncthreads();
    uint8_t *out = slots + b * (uint64_t)DEFL_SLOT;
    const uint32_t payload = stored ? (uint32_t)n + 5 : cbytes;
    if (stored) {
        for (int i = t; i < (n >> 2); i += DEFL_THREADS) d_st32(out + 23 + 4 * i, in4[i + (i >> 4)]);
        for (int i = (n & ~3) + t; i < n; i += DEFL_THREADS) out[23 + i] = DEFL_INB(in, i);
        if (t == 0) { out[18] = 1; d_st16(out + 19, (uint32_t)n); d_st16(out + 21, (uint32_t)(~n & 0xffff)); }
    } else {
        // payload at out + 18: 2 bytes past a dword boundary
        for (uint32_t i = t; i < (cbytes + 3) >> 2; i += DEFL_THREADS) {
            const uint32_t v = out4[i];
            if (4 * i + 4 <= cbytes) { d_st16(out + 18 + 4 * i, v & 0xffffu); d_st16(out + 20 + 4 * i, v >> 16); }
            else for (uint32_t k = 0; 4 * i + k < cbytes; ++k) out[18 + 4 * i + k] = (uint8_t)(v >> (8 * k));
        }
    }
    if (t == 0) {
        uint32_t crc = 0; for (int i = 0; i < 16; ++i) crc ^= wsum[i];
        const uint8_t hdr[16] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0 };
        for (int i = 0; i < 16; ++i) out[i] = hdr[i];
        d_st16(out + 16, payload + 26 - 1);
        d_st32(out + 18 + payload, crc); d_st32(out + 22 + payload, (uint32_t)n);
        csize[b] = payload + 26;
    }
}

__global__ void __launch_bounds__(256) k_bgzf_compact(const uint8_t *__restrict__ slots, const uint32_t *__restrict__ csize, const uint64_t *__restrict__ coff, uint8_t *__restrict__ dst, uint32_t blk0)
{
    const uint64_t b = (uint64_t)blockIdx.x + blk0;
    const uint32_t n = csize[b];
    const uint32_t *s4 = (const uint32_t*)(slots + b * (uint64_t)DEFL_SLOT);
    uint8_t *d = dst + coff[b];
    for (uint32_t i = threadIdx.x; i < n >> 2; i += 256) d_st32(d + 4 * i, s4[i]);
    for (uint32_t i = (n & ~3u) + threadIdx.x; i < n; i += 256) d[i] = slots[b * (uint64_t)DEFL_SLOT + i];
}
__global__ void __launch_bounds__(256) k_widen_u32(const uint32_t *__restrict__ in, int32_t n, uint64_t *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i]; else if (i == n) out[i] = 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
static void crc_tabs_make(CrcTabs &T)
{
    for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1; T.byte_tab[i] = c; }
    auto mul = [](uint32_t a, uint32_t b) { uint32_t p = 0; for (uint32_t m = 0x80000000u; m; m >>= 1) { if (a & m) p ^= b; b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1; } return p; };
    // x^(8*255): start from x^0 = 0x80000000 (reflected) and multiply by x 8*255 times
    uint32_t x1 = 0x80000000u;
    for (int i = 0; i < 8 * 255; ++i) x1 = (x1 & 1u) ? (x1 >> 1) ^ 0xEDB88320u : x1 >> 1;
    T.xpow255[0] = 0x80000000u;
    for (int k = 1; k < 256; ++k) T.xpow255[k] = mul(T.xpow255[k - 1], x1);
    uint32_t x64 = 0x80000000u;
    for (int i = 0; i < 8 * 64; ++i) x64 = (x64 & 1u) ? (x64 >> 1) ^ 0xEDB88320u : x64 >> 1;
    T.xpow64[0] = 0x80000000u;
    for (int k = 1; k < 1024; ++k) T.xpow64[k] = mul(T.xpow64[k - 1], x64);
}


// ---- Huffman tables of the device deflate (host) ---------------------------------------------------------------------
// code lengths limited to `maxlen` bits: Huffman depths, then the overflow moved down the lengths (the method of miniz /
// zlib's gen_bitlen: fold codes longer than the limit into it, repair the Kraft sum, hand the longest codes to the rarest symbols)
static void huff_lengths(const uint32_t *freq, int n, int maxlen, uint8_t *lens)
{
    std::vector<int> sym;
    for (int i = 0; i < n; ++i) { lens[i] = 0; if (freq[i]) sym.push_back(i); }
    if (sym.empty()) return;
    if (sym.size() == 1) { lens[sym[0]] = 1; return; }
    std::stable_sort(sym.begin(), sym.end(), [&](int a, int b) { return freq[a] < freq[b]; });
    const int m = (int)sym.size();
    // two-queue Huffman on the sorted leaves
    std::vector<uint64_t> w(2 * m); std::vector<int> parent(2 * m, -1);
    for (int i = 0; i < m; ++i) w[i] = freq[sym[i]];
    int leaf = 0, inode = m, next = m;
    auto take = [&]() { if (leaf < m && (inode >= next || w[leaf] <= w[inode])) return leaf++; return inode++; };
    while (next < 2 * m - 1) { const int a = take(), b = take(); w[next] = w[a] + w[b]; parent[a] = parent[b] = next; ++next; }
    std::vector<int> depth(2 * m, 0);
    for (int i = 2 * m - 3; i >= 0; --i) depth[i] = depth[parent[i]] + 1;
    std::vector<int> cnt(64, 0);
    for (int i = 0; i < m; ++i) ++cnt[depth[i] < 63 ? depth[i] : 63];
    for (int l = 63; l > maxlen; --l) { cnt[maxlen] += cnt[l]; cnt[l] = 0; }
    uint64_t total = 0;
    for (int l = maxlen; l >= 1; --l) total += (uint64_t)cnt[l] << (maxlen - l);
    while (total != (1ull << maxlen)) {
        --cnt[maxlen];
        for (int l = maxlen - 1; l >= 1; --l) if (cnt[l]) { --cnt[l]; cnt[l + 1] += 2; break; }
        --total;
    }
    int k = 0;        // rarest symbols first: longest codes
    for (int l = maxlen; l >= 1; --l) for (int c = 0; c < cnt[l]; ++c) lens[sym[k++]] = (uint8_t)l;
}
static void huff_codes(const uint8_t *lens, int n, uint32_t *codes_rev)
{
    int bl[17] = {0}; uint32_t nc[17] = {0};
    for (int i = 0; i < n; ++i) ++bl[lens[i]];
    bl[0] = 0;
    uint32_t code = 0;
    for (int l = 1; l <= 16; ++l) { code = (code + bl[l - 1]) << 1; nc[l] = code; }
    for (int i = 0; i < n; ++i) {
        const int l = lens[i]; uint32_t c = 0;
        if (l) { uint32_t v = nc[l]++; for (int b = 0; b < l; ++b) c |= ((v >> b) & 1u) << (l - 1 - b); }
        codes_rev[i] = c;
    }
}
struct HostBits { std::vector<uint32_t> w; uint32_t n = 0; void put(uint32_t v, int len) { for (int i = 0; i < len; ++i) { if ((n >> 5) >= w.size()) w.push_back(0); w[n >> 5] |= ((v >> i) & 1u) << (n & 31); ++n; } } };
// the header of a dynamic block (RFC 1951 3.2.7) for literal/length lengths ll[286] and ONE distance code of length 1
static void deflate_dyn_header(const uint8_t *ll, HostBits &B)
{
    std::vector<uint8_t> all(ll, ll + 286); all.push_back(1);
    struct Sy { uint8_t s, ebits, eval; };
    std::vector<Sy> rl;
    for (size_t i = 0; i < all.size(); ) {
        size_t j = i; while (j < all.size() && all[j] == all[i]) ++j;
        size_t run = j - i; const uint8_t v = all[i];
        if (v == 0) {
            while (run >= 11) { size_t r = std::min<size_t>(run, 138); rl.push_back(Sy{18, 7, (uint8_t)(r - 11)}); run -= r; }
            if (run >= 3) { rl.push_back(Sy{17, 3, (uint8_t)(run - 3)}); run = 0; }
            while (run--) rl.push_back(Sy{0, 0, 0});
        } else {
            rl.push_back(Sy{v, 0, 0}); --run;
            while (run >= 3) { size_t r = std::min<size_t>(run, 6); rl.push_back(Sy{16, 2, (uint8_t)(r - 3)}); run -= r; }
            while (run--) rl.push_back(Sy{v, 0, 0});
        }
        i = j;
    }
    uint32_t f[19] = {0}; for (auto &x : rl) ++f[x.s];
    uint8_t cl[19]; huff_lengths(f, 19, 7, cl);
    uint32_t cc[19]; huff_codes(cl, 19, cc);
    static const int ord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19; while (hclen > 4 && cl[ord[hclen - 1]] == 0) --hclen;
    B.put(286 - 257, 5); B.put(0, 5); B.put((uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; ++i) B.put(cl[ord[i]], 3);
    for (auto &x : rl) { B.put(cc[x.s], cl[x.s]); if (x.ebits) B.put(x.eval, x.ebits); }
}
static void defl_len_syms(uint16_t *len_sym)
{
    static const int base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const int eb[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    for (int L = 0; L < 259; ++L) {
        if (L < 3) { len_sym[L] = 0; continue; }
        int c = 28; while (base[c] > L) --c;
        if (L == 258) c = 28;
        len_sym[L] = (uint16_t)(c | eb[c] << 5 | (L - base[c]) << 8);
    }
}
// histogram[class][288] -> tables.  Every literal, the end-of-block symbol and every length symbol keeps a code (count >= 1):
// any byte may turn up in any class (fields shorter than DEFL_MINSEG ride on the table in force).
static int defl_tables_from_hist(const uint32_t *hist, DeflTabs &T)
{
    for (int c = 0; c < DEFL_NCLS; ++c) {
        uint32_t f[286];
        for (int i = 0; i < 286; ++i) f[i] = hist[c * 288 + i] + 1;
        uint8_t ll[286]; huff_lengths(f, 286, 15, ll);
        uint32_t codes[286]; huff_codes(ll, 286, codes);
        for (int i = 0; i < 288; ++i) T.lit[c][i] = i < 286 ? (codes[i] | (uint32_t)ll[i] << 16) : 0;
        HostBits B; deflate_dyn_header(ll, B);
        if (B.w.size() > 96) return TELR_E_RANGE;
        T.hdr_bits[c] = B.n;
        memset(T.hdr[c], 0, sizeof(T.hdr[c]));
        memcpy(T.hdr[c], B.w.data(), B.w.size() * 4);
    }
    return TELR_OK;
}
// CPU tap for tests/test_deflate_tables.py: code lengths (length-limited, complete) and the dynamic-block header bit string
extern "C" int telr_debug_huff(const uint32_t *freq, int32_t n, int32_t maxlen, uint8_t *lens_out)
{
    if (!freq || !lens_out || n < 1 || n > 288 || maxlen < 1 || maxlen > 15) return TELR_E_ARG;
    huff_lengths(freq, n, maxlen, lens_out);
    return TELR_OK;
}
// a whole deflate stream for `src` with ONE class table built from its own histogram, literals + distance-1 matches cut at
// pieces of 64 bytes: the host restatement of what k_bgzf_deflate emits for a one-segment block (zlib must inflate it to src)
extern "C" int telr_debug_deflate_host(const uint8_t *src, int32_t n, uint8_t *out, int32_t cap, int32_t *out_len)
{
    if (!src || n < 0 || !out || !out_len) return TELR_E_ARG;
    uint16_t len_sym[259]; defl_len_syms(len_sym);
    std::vector<uint32_t> hist(DEFL_NCLS * 288, 0);
    auto walk = [&](auto &&lit, auto &&match) {
        for (int p0 = 0; p0 < n; p0 += DEFL_PIECE) {
            const int p1 = std::min(n, p0 + DEFL_PIECE);
            int i = p0; int prev = i > 0 ? src[i - 1] : 256;
            while (i < p1) {
                const int b = src[i];
                if (b == prev) { int L = 1; while (i + L < p1 && src[i + L] == b) ++L; if (L >= 3) { match(L); i += L; continue; } }
                lit(b); prev = b; ++i;
            }
        }
    };
    walk([&](int b) { ++hist[b]; }, [&](int L) { ++hist[257 + (len_sym[L] & 31)]; });
    ++hist[256];
    DeflTabs T; memset(&T, 0, sizeof(T));
    int rc = defl_tables_from_hist(hist.data(), T);
    if (rc != TELR_OK) return rc;
    HostBits B;
    B.put(1u | 2u << 1, 3);
    for (uint32_t i = 0; i < T.hdr_bits[0]; ++i) B.put((T.hdr[0][i >> 5] >> (i & 31)) & 1u, 1);
    auto sym = [&](int s) { B.put(T.lit[0][s] & 0xffffu, (int)(T.lit[0][s] >> 16)); };
    walk([&](int b) { sym(b); }, [&](int L) { const uint32_t ls = len_sym[L]; sym(257 + (int)(ls & 31)); if (ls >> 5 & 7) B.put(ls >> 8, (int)(ls >> 5 & 7)); B.put(0, 1); });
    sym(256);
    const int nb = (int)((B.n + 7) / 8);
    if (nb > cap) return TELR_E_RANGE;
    memcpy(out, B.w.data(), (size_t)nb);
    *out_len = nb;
    return TELR_OK;
}

struct BamTimes { float ms[8]; };        // upload, scan+size, sort+scan, write, bgzf, d2h+file, bai (host, overlapped), total
static BamTimes g_bam_times;
extern "C" int telr_debug_bam_ms(float *out) { if (!out) return TELR_E_ARG; memcpy(out, g_bam_times.ms, sizeof(g_bam_times.ms)); return TELR_OK; }

// the .bai of a coordinate-sorted record sequence: alns in sorted order through `order`, uncompressed start of every
// record (sorted order, [nrec + 1]) and the file offset of every BGZF block ([nblk + 1])
// Two steps, so that the long one needs nothing the coder produces: bai_build lays the whole index out with UNCOMPRESSED stream
// offsets in the place of virtual file offsets (same order, and "same BGZF block" is "same 65,280-byte piece of the stream")
// and notes where they stand; bai_finish, once the blocks' file offsets are known, rewrites those fields.
static void bai_finish(std::string &bai, const std::vector<size_t> &fix, const uint64_t *coff, size_t nblk)
{
    auto voff = [&](uint64_t u) { size_t b = (size_t)(u / BAM_BLK); if (b >= nblk) return (uint64_t)(coff[nblk] << 16); return (uint64_t)(coff[b] << 16 | (u - (uint64_t)b * BAM_BLK)); };
    for (size_t p : fix) { uint64_t u; memcpy(&u, &bai[p], 8); const uint64_t v = voff(u); memcpy(&bai[p], &v, 8); }
}
static void bai_build(const std::vector<telr_aln> &recs, const uint32_t *order, size_t nrec, size_t n_unmapped, const uint64_t *ustart,
                      int32_t n_targets, const int32_t *t_len, std::string &bai, std::vector<size_t> &fix)
{
    auto put32 = [&](uint32_t v) { bai.append((const char*)&v, 4); };
    auto put_off = [&](std::string &dst, size_t base, uint64_t u) { fix.push_back(base + dst.size()); dst.append((const char*)&u, 8); };      // base: where dst will start inside bai
    auto voff = [&](uint64_t u) { return u; };
    auto blk_of = [&](uint64_t u) { return u / BAM_BLK; };
    fix.clear();
    bai = "BAI\1"; put32((uint32_t)n_targets);
    size_t i = 0;
    const size_t n_mapped = nrec - n_unmapped;
    struct Ch { uint32_t bin; uint64_t vb, ve; };
    std::vector<Ch> chs;
    for (int t = 0; t < n_targets; ++t) {
        chs.clear();
        const int n_lin = (t_len[t] >> 14) + 1;
        std::vector<uint64_t> lin(n_lin, 0);
        int max_lin = 0;
        uint64_t ref_beg = 0, ref_end = 0, n_map = 0;
        bool any = false;
        while (i < n_mapped && recs[order[i]].tid == t) {
            const telr_aln &a = recs[order[i]];
            const uint64_t vb = voff(ustart[i]), ve = voff(ustart[i + 1]);
            const int e = a.te > a.ts ? a.te : a.ts + 1;
            chs.push_back(Ch{ (uint32_t)reg2bin(a.ts, e), vb, ve });
            const int w0 = a.ts >> 14, w1 = (e - 1) >> 14;
            for (int wv = w0; wv <= w1 && wv < n_lin; ++wv) { if (lin[wv] == 0 || vb < lin[wv]) lin[wv] = vb; if (wv + 1 > max_lin) max_lin = wv + 1; }
            if (!any) { ref_beg = vb; any = true; }
            ref_end = ve; ++n_map; ++i;
        }
        std::stable_sort(chs.begin(), chs.end(), [](const Ch &x, const Ch &y) { return x.bin < y.bin; });
        // bins in ascending order, chunks of a bin merged while they end and start in the same BGZF block
        std::string body; uint32_t nbin = 0;
        for (size_t c0 = 0; c0 < chs.size(); ) {
            size_t c1 = c0; std::vector<std::pair<uint64_t, uint64_t>> ch;
            while (c1 < chs.size() && chs[c1].bin == chs[c0].bin) {
                if (!ch.empty() && blk_of(ch.back().second) == blk_of(chs[c1].vb)) ch.back().second = chs[c1].ve; else ch.push_back(std::make_pair(chs[c1].vb, chs[c1].ve));
                ++c1;
            }
            uint32_t bin = chs[c0].bin, nc = (uint32_t)ch.size();
            body.append((const char*)&bin, 4); body.append((const char*)&nc, 4);
            for (auto &c : ch) { put_off(body, bai.size() + 4, c.first); put_off(body, bai.size() + 4, c.second); }      // body goes in behind the 4-byte bin count
            ++nbin; c0 = c1;
        }
        put32(nbin + (any ? 1 : 0));
        bai += body;
        if (any) {   // samtools' metadata pseudo-bin 37450
            put32(37450u); put32(2u);
            put_off(bai, 0, ref_beg); put_off(bai, 0, ref_end);
            uint64_t zero = 0; bai.append((const char*)&n_map, 8); bai.append((const char*)&zero, 8);
        }
        for (int wv = 1; wv < max_lin; ++wv) if (lin[wv] == 0) lin[wv] = lin[wv - 1];
        put32((uint32_t)max_lin);
        // (a window before the first record of the reference keeps 0 = "from the start of the file", as it does with virtual offsets)
        for (int wv = 0; wv < max_lin; ++wv) { if (lin[wv]) put_off(bai, 0, lin[wv]); else bai.append((const char*)&lin[wv], 8); }
    }
    uint64_t n_no_coor = n_unmapped;
    bai.append((const char*)&n_no_coor, 8);
}

static std::string bam_header(int32_t n_targets, const char *const *tnames, const int32_t *t_len, const char *rg_id, const char *rg_sm, const char *rg_lb, const char *pg_line)
{
    std::string head, text = "@HD\tVN:1.6\tSO:coordinate\n";
    char b[512];
    for (int t = 0; t < n_targets; ++t) { snprintf(b, sizeof(b), "@SQ\tSN:%s\tLN:%d\n", tnames[t], t_len[t]); text += b; }
    if (rg_id) { snprintf(b, sizeof(b), "@RG\tID:%s\tSM:%s\tLB:%s\n", rg_id, rg_sm ? rg_sm : rg_id, rg_lb ? rg_lb : "lib"); text += b; }
    text += "@PG\tID:telr_amd\tPN:telr_amd\tVN:0.1.0\tCL:"; text += pg_line ? pg_line : "telr_map"; text += "\n";
    auto put32 = [&](uint32_t v) { head.append((const char*)&v, 4); };
    head += "BAM\1"; put32((uint32_t)text.size()); head += text; put32((uint32_t)n_targets);
    for (int t = 0; t < n_targets; ++t) { uint32_t ln = (uint32_t)strlen(tnames[t]) + 1; put32(ln); head.append(tnames[t], ln); put32((uint32_t)t_len[t]); }
    return head;
}

// ---- the output file, prepared ahead -----------------------------------------------------------------------------------
// What a tmpfs file costs is the allocation of its pages and their way into the writer's page table (tools/ubench/shm_io.hip on
// the MI355X box: pwrite into a fresh file 6.4-6.7 GB/s with one thread and LESS with more -- writes to one inode serialise --,
// posix_fallocate 17-19 GB/s, pwrite over allocated pages 9.7 GB/s, memcpy through a mapping of the allocated file 12-15 GB/s
// with 4-16 threads but 80-200 GB/s once the mapping is populated; DMA straight into a registered mapping runs at 50-57 GB/s
// but registering costs 0.47 s per 3.9 GB, pages present or not, and stalls every other HIP call of the process meanwhile:
// measured, rejected).  telr_bam_prepare() creates and maps the file at once and starts a thread that allocates and pre-faults
// it block by block, front to back, while the caller is still mapping reads; the writer moves the finished image through the
// pinned ring into the prepared prefix of the mapping and cuts the file to its real length; the mapping is taken apart by a
// background thread (bam_sink_drop).
// events of one writer call: destroyed on every way out
struct EvBag { std::vector<hipEvent_t> v; ~EvBag() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); } };
struct BamSink {
    std::string path; int fd = -1; uint8_t *map = nullptr; size_t bytes = 0;
    std::thread th; float ms_alloc = 0, ms_map = 0;
    // The file becomes usable front to back, block by block: a block is allocated (tmpfs takes one allocator at a time: 17-19 GB/s)
    // and then pre-faulted into the mapping by four threads (MADV_POPULATE_WRITE, ~38 GB/s): a copy into a populated mapping runs
    // at 80-200 GB/s, into an allocated but unmapped one at 12-15 GB/s -- the page-fault path (profiles/r03_shm_io_populate.txt).
    // The two steps alternate: run side by side they contend for the file's page tree (allocation 230 -> 610 ms).
    static constexpr size_t SLICE = 64u << 20;
    std::mutex mu; std::condition_variable cv; size_t n_slices = 0, ready = 0, limit = ~(size_t)0; bool stop = false, over = false;
    bool alloc_failed = false;           // posix_fallocate refused a block (ENOSPC / EDQUOT): nothing past the prepared prefix is backed by pages
    // wait until the first `end` bytes are allocated and mapped, or the sink has given up on them; -> the length of the prefix
    // that IS allocated.  Only that prefix may be written through the mapping: a store into a hole of a full file system is a
    // SIGBUS, where pwrite returns ENOSPC -- the writer takes pwrite for everything beyond it.
    size_t wait_ready(size_t end)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return over || ready * SLICE >= std::min(std::min(end, bytes), limit); });
        return std::min(ready * SLICE, bytes);
    }
    // the writer's running estimate of the file's length: blocks past it are not prepared (cutting prepared pages off the end
    // of the file costs ~0.18 s per GB: the pages go back one by one)
    void set_limit(size_t b)
    {
        { std::lock_guard<std::mutex> lk(mu); limit = b; }
        cv.notify_all();
    }
};
static int g_bam_twin = 0;          // the last writer call found the CIGARs on the device
extern "C" int telr_debug_bam_twin(void) { return g_bam_twin; }
static float g_sink_ms[12];      // allocate + populate, map (prepare threads); wait for them (writer); 1 = the mapping was used; streaming: wait for the coder, wait for the DMA, host copies / pwrite, wait for a ring slot, the streaming loop as a whole, stopping the prepare thread, cutting the file
static std::mutex g_rel_mu; static std::condition_variable g_rel_cv; static int g_rel_pending = 0;
static void bam_sink_drop(telr_ctx *ctx)
{
    BamSink *k = ctx->bam_sink;
    if (!k) return;
    { std::lock_guard<std::mutex> lk(k->mu); k->stop = true; }
    k->cv.notify_all();
    if (k->th.joinable()) k->th.join();
    // taking a populated 4-GB mapping apart costs ~200 ms of page-table work: not on the caller's clock (the file is complete
    // and cut to its length by now; the thread owns nothing but the mapping and the descriptor)
    uint8_t *map = k->map; const size_t bytes = k->bytes; const int fd = k->fd;
    if (map && bytes >= ((size_t)256 << 20)) {
        { std::lock_guard<std::mutex> lk(g_rel_mu); ++g_rel_pending; }
        std::thread([map, bytes, fd] {
            unmap_in_pieces(map, bytes); if (fd >= 0) close(fd);
            { std::lock_guard<std::mutex> lk(g_rel_mu); --g_rel_pending; }
            g_rel_cv.notify_all();
        }).detach();
    } else { if (map) munmap(map, bytes); if (fd >= 0) close(fd); }
    delete k; ctx->bam_sink = nullptr;
}
// wait for the mappings of earlier output files to be taken apart (a process that writes one BAM never needs this; one that
// writes several back to back would find its next telr_bam_prepare waiting for the address-space lock the release holds)
extern "C" int telr_bam_release_wait(void)
{
    std::unique_lock<std::mutex> lk(g_rel_mu);
    g_rel_cv.wait(lk, [] { return g_rel_pending == 0; });
    return TELR_OK;
}
extern "C" int telr_bam_discard(telr_ctx *ctx)
{
    if (!ctx) return TELR_E_ARG;
    bam_sink_drop(ctx);
    return TELR_OK;
}
extern "C" int telr_bam_prepare(telr_ctx *ctx, const char *bam_path, int64_t est_bytes)
{
    if (!ctx || !bam_path || est_bytes <= 0) return TELR_E_ARG;
    bam_sink_drop(ctx);
    BamSink *k = new BamSink();
    k->path = bam_path; k->bytes = ((size_t)est_bytes + 4095) & ~(size_t)4095;
    k->n_slices = (k->bytes + BamSink::SLICE - 1) / BamSink::SLICE;          // the last block may be a short one
    // file and mapping first (both are instant), so that the writer can use whatever part is ready when it arrives
    auto t0 = std::chrono::steady_clock::now();
    k->fd = open(k->path.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644);
    if (k->fd >= 0 && ftruncate(k->fd, (off_t)k->bytes) == 0) {
        void *m = mmap(nullptr, k->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, k->fd, 0);
        if (m != MAP_FAILED) k->map = (uint8_t*)m;
    }
    k->ms_map = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (k->fd < 0 || !k->map) { k->over = true; ctx->bam_sink = k; return TELR_OK; }      // the writer falls back to plain streaming
    static const bool no_populate = ab_on("bam_no_populate");
    k->th = std::thread([k] {
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < k->n_slices; ++i) {
            { std::unique_lock<std::mutex> lk(k->mu); k->cv.wait(lk, [&] { return k->stop || i * BamSink::SLICE < k->limit; }); if (k->stop) break; }
            const size_t len = std::min(BamSink::SLICE, k->bytes - i * BamSink::SLICE);
            if (posix_fallocate(k->fd, (off_t)(i * BamSink::SLICE), (off_t)len) != 0) { std::lock_guard<std::mutex> lk(k->mu); k->alloc_failed = true; break; }
            if (!no_populate) {
                const int P = 4; const size_t q = (len / P + 4095) & ~(size_t)4095;
                std::thread pt[P];
                for (int t = 0; t < P; ++t) pt[t] = std::thread([k, i, t, q, len] {
                    const size_t lo = std::min(len, (size_t)t * q), n = std::min(len - lo, q);
                    if (!n) return;
                    uint8_t *p = k->map + i * BamSink::SLICE + lo;
                    if (madvise(p, n, 23 /* MADV_POPULATE_WRITE, Linux 5.14 */) != 0 && errno == EINVAL)
                        for (size_t o = 0; o < n; o += 4096) ((volatile uint8_t*)p)[o] = 0;       // older kernels: touch the pages
                });
                for (int t = 0; t < P; ++t) pt[t].join();
            }
            { std::lock_guard<std::mutex> lk(k->mu); k->ready = i + 1; }
            k->cv.notify_all();
        }
        { std::lock_guard<std::mutex> lk(k->mu); k->over = true; }
        k->cv.notify_all();
        k->ms_alloc = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    });
    ctx->bam_sink = k;
    return TELR_OK;
}

// Device image -> file, through a ring of pinned chunks: the DMA of chunk c+1 runs while the writer thread puts chunk c into
// the file.  The image may still be GROWING while it streams out (the deflate groups of telr_write_bam_dev): `prog` tells how
// many of its leading bytes are final; a null `prog` means all `bytes` are.  `tail` (host bytes) is appended.  fd >= 0: an
// open file whose pages may already exist (it is cut to the final length), else `path` is created.  map_dst / map_bytes: a
// mapping of that (allocated) file: chunks inside it are copied by the host pool instead of pwrite.
struct StreamProgress { std::mutex mu; std::condition_variable cv; uint64_t ready = 0; int state = 0; uint64_t total = 0; };    // state 0 producing, 1 done (total valid), -1 failed
static void progress_set(StreamProgress *p, uint64_t ready, int state, uint64_t total)
{
    { std::lock_guard<std::mutex> lk(p->mu); if (ready > p->ready) p->ready = ready; if (state) { p->state = state; p->total = total; } }
    p->cv.notify_all();
}
static int stream_to_file(telr_ctx *ctx, const uint8_t *d_img, uint64_t bytes, StreamProgress *prog, const void *tail, size_t tail_bytes, const char *path,
                          int fd_open = -1, uint8_t *map_dst = nullptr, size_t map_bytes = 0, uint64_t *total_out = nullptr, BamSink *sink = nullptr, float *sink_wait_ms = nullptr,
                          uint64_t file_off = 0, bool keep_size = false)      // file_off: the image goes to this offset of the file (map_dst = the mapping of that place); keep_size: do not cut the file
{
    const size_t CH = 32u << 20; const int R = 8;
    uint8_t *ring; TRY(ctx_hbuf_t(ctx, "bam_ring", CH * R, &ring));
    int fd = fd_open >= 0 ? fd_open : open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
    if (fd < 0) { ctx->err = std::string("cannot create ") + path; return TELR_E_ARG; }
    EvBag ring_ev; ring_ev.v.assign(R, nullptr);
    hipEvent_t *ev = ring_ev.v.data();
    for (int i = 0; i < R; ++i) if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { if (fd_open < 0) close(fd); return TELR_E_HIP; }
    // size of chunk c once it is final (0: the image ended before it; -1: failed; -2: not known yet and !block)
    auto chunk_bytes = [&](size_t c, bool block) -> int64_t {
        if (!prog) { const uint64_t o = (uint64_t)c * CH; return o >= bytes ? 0 : (int64_t)std::min<uint64_t>(CH, bytes - o); }
        std::unique_lock<std::mutex> lk(prog->mu);
        auto known = [&] { return prog->state != 0 || prog->ready >= (uint64_t)(c + 1) * CH; };
        if (!known()) { if (!block) return -2; prog->cv.wait(lk, known); }
        if (prog->state < 0) return -1;
        const uint64_t end = prog->state == 1 ? prog->total : prog->ready, o = (uint64_t)c * CH;
        return o >= end ? 0 : (int64_t)std::min<uint64_t>(CH, end - o);
    };
    // One writer thread: a single pwrite stream is what a tmpfs file takes fastest (see above); into a mapping, eight copiers
    std::mutex mu; std::condition_variable cv; size_t copied = 0, written = 0; bool fail = false, last = false; std::vector<size_t> csz;
    std::thread writer([&] {
        for (size_t c = 0;; ++c) {
            size_t n;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return copied > c || fail || last; }); if (fail) return; if (copied <= c) return; n = csz[c]; }
            const uint8_t *src = ring + (c % R) * CH;
            const auto tw0 = std::chrono::steady_clock::now();
            struct Acc { std::chrono::steady_clock::time_point t0; ~Acc() { g_sink_ms[6] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } acc{tw0};
            bool through_map = map_dst && (uint64_t)c * CH + n <= map_bytes;
            if (through_map && sink) {
                const auto t0 = std::chrono::steady_clock::now();
                through_map = sink->wait_ready((size_t)c * CH + n) >= (size_t)c * CH + n;      // pages exist for the whole chunk
                if (sink_wait_ms) *sink_wait_ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
            if (through_map) {
                const int NT = std::max(4, std::min(16, host_threads())); const size_t piece = (n + NT - 1) / NT;      // 12.4 / 12.8 / 14.6 GB/s with 4 / 8 / 16 copiers (shm_io)
                uint8_t *dst = map_dst + (uint64_t)c * CH;
                HostPool::get().run(NT, [&](int i) { const size_t o = (size_t)i * piece; if (o < n) memcpy(dst + o, src + o, std::min(piece, n - o)); });
            } else {
                size_t done = 0;
                while (done < n) { ssize_t w = pwrite(fd, src + done, n - done, (off_t)(file_off + (uint64_t)c * CH + done)); if (w <= 0) { std::lock_guard<std::mutex> lk(mu); fail = true; cv.notify_all(); return; } done += (size_t)w; }
            }
            { std::lock_guard<std::mutex> lk(mu); written = c + 1; }
            cv.notify_all();
        }
    });
    int rc = TELR_OK;
    size_t issued = 0, landed = 0; bool ended = false; uint64_t total = 0;
    while (rc == TELR_OK) {
        // issue every chunk that is final and has a free ring slot (never blocks on the producer while copies are in flight)
        bool slot_wait = false;
        while (!ended && issued < landed + (size_t)R - 1) {
            { std::lock_guard<std::mutex> lk(mu); if (fail) break; if (issued >= written + (size_t)R) { slot_wait = true; break; } }
            const auto tp0 = std::chrono::steady_clock::now();
            const int64_t n = chunk_bytes(issued, issued == landed);        // nothing in flight: wait for the producer
            g_sink_ms[4] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tp0).count();
            if (n == -2) break;
            if (n < 0) { rc = TELR_E_HIP; break; }
            if (n == 0) { ended = true; break; }
            if (hipMemcpyAsync(ring + (issued % R) * CH, d_img + (uint64_t)issued * CH, (size_t)n, hipMemcpyDeviceToHost, ctx->copy_stream) != hipSuccess ||
                hipEventRecord(ev[issued % R], ctx->copy_stream) != hipSuccess) { rc = TELR_E_HIP; break; }
            { std::lock_guard<std::mutex> lk(mu); csz.push_back((size_t)n); }
            total += (uint64_t)n; ++issued;
            if ((size_t)n < CH) ended = true;
        }
        { std::lock_guard<std::mutex> lk(mu); if (fail) break; }
        if (rc != TELR_OK) break;
        if (landed < issued) {
            const auto te0 = std::chrono::steady_clock::now();
            if (hipEventSynchronize(ev[landed % R]) != hipSuccess) { rc = TELR_E_HIP; break; }
            g_sink_ms[5] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - te0).count();
            ++landed;
            { std::lock_guard<std::mutex> lk(mu); copied = landed; }
            cv.notify_all();
        } else if (ended) break;
        else if (slot_wait) {
            const auto ts0 = std::chrono::steady_clock::now();
            std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return issued < written + (size_t)R || fail; });
            g_sink_ms[7] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - ts0).count();
        }
    }
    { std::lock_guard<std::mutex> lk(mu); if (rc != TELR_OK) fail = true; last = true; }
    cv.notify_all();
    writer.join();
    if (fail && rc == TELR_OK) { ctx->err = std::string("write to ") + path + " failed"; rc = TELR_E_ARG; }
    if (rc == TELR_OK && tail_bytes) {
        const bool in_map = map_dst && total + tail_bytes <= map_bytes && (!sink || sink->wait_ready((size_t)(total + tail_bytes)) >= total + tail_bytes);
        if (in_map) memcpy(map_dst + total, tail, tail_bytes);
        else if (pwrite(fd, tail, tail_bytes, (off_t)(file_off + total)) != (ssize_t)tail_bytes) { ctx->err = std::string("write to ") + path + " failed"; rc = TELR_E_ARG; }
    }
    if (rc == TELR_OK && fd_open >= 0 && !sink && !keep_size && ftruncate(fd, (off_t)(total + tail_bytes)) != 0) rc = TELR_E_ARG;      // with a sink: cut by the caller, once its threads have stopped
    if (fd_open < 0) close(fd);
    if (total_out) *total_out = total;
    return rc;
}
// the same through a prepared sink (telr_bam_prepare); plain streaming when there is none for this path
static int sink_to_file(telr_ctx *ctx, const uint8_t *d_img, uint64_t bytes, StreamProgress *prog, const void *tail, size_t tail_bytes, const char *path)
{
    BamSink *k = ctx->bam_sink;
    memset(g_sink_ms, 0, sizeof(g_sink_ms));
    if (!k || k->path != path) { if (k) bam_sink_drop(ctx); return stream_to_file(ctx, d_img, bytes, prog, tail, tail_bytes, path); }
    if (k->fd < 0 || !k->map) { bam_sink_drop(ctx); return stream_to_file(ctx, d_img, bytes, prog, tail, tail_bytes, path); }
    uint64_t total = 0;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](int i) { const auto t1 = std::chrono::steady_clock::now(); g_sink_ms[i] = std::chrono::duration<float, std::milli>(t1 - t0).count(); t0 = t1; };
    int rc = stream_to_file(ctx, d_img, bytes, prog, tail, tail_bytes, path, k->fd, k->map, k->bytes, &total, k, &g_sink_ms[2]);
    lap(8);
    { std::lock_guard<std::mutex> lk(k->mu); k->stop = true; }      // slices past the end of the image are not needed
    k->cv.notify_all();
    if (k->th.joinable()) k->th.join();
    lap(9);
    if (rc == TELR_OK && ftruncate(k->fd, (off_t)(total + tail_bytes)) != 0) rc = TELR_E_ARG;
    lap(10);
    g_sink_ms[0] = k->ms_alloc; g_sink_ms[1] = k->ms_map;
    g_sink_ms[3] = total + tail_bytes <= k->bytes ? 1.f : 0.f;
    return rc;          // the caller drops the sink once its other host threads are done: taking the mapping apart holds the address-space lock they allocate under
}
extern "C" int telr_debug_bam_sink_ms(float *out) { if (!out) return TELR_E_ARG; memcpy(out, g_sink_ms, sizeof(g_sink_ms)); return TELR_OK; }

// The mapping scratch of a context is grow-only (a 30x read set leaves 150-250 GB of it behind) and the writer needs about
// 3.3 bytes of HBM per byte of uncompressed BAM: when an allocation fails, the mapping scratch of the context and of its worker
// contexts is given back (the next telr_map call sizes it again) and the writer runs once more.
static uint64_t g_bam_need = 0;          // device bytes the writer was about to use when an allocation failed
static void ctx_collect_scratch(telr_ctx *ctx, std::vector<DBuf*> &v)
{
    for (auto &kv : ctx->bufs) if (kv.first.compare(0, 4, "bam_") != 0 && kv.second.p) v.push_back(&kv.second);
    if (ctx->slot1) ctx_collect_scratch(ctx->slot1, v);
}
// largest buffers first, until `need` bytes are free (the next telr_map call allocates what it misses again: ~25 ms per GB)
static void ctx_release_map_scratch(telr_ctx *ctx, uint64_t need)
{
    (void)hipDeviceSynchronize();
    std::vector<DBuf*> v; ctx_collect_scratch(ctx, v);
    std::sort(v.begin(), v.end(), [](const DBuf *a, const DBuf *b) { return a->bytes > b->bytes; });
    for (DBuf *d : v) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr >= need + (need >> 3)) break;
        (void)hipFree(d->p); d->p = nullptr; d->bytes = 0;
    }
}
// everything the context holds for its calls (mapping scratch of all its slots and workers, the writer's buffers): the next call
// sizes it again (~2 ms per GB).  For a context that shares the device with another one of the process (telr_init_background).
extern "C" int telr_release_scratch(telr_ctx *ctx)
{
    if (!ctx) return TELR_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    (void)hipDeviceSynchronize();
    std::vector<DBuf*> v; ctx_collect_scratch(ctx, v);
    for (auto &kv : ctx->bufs) if (kv.first.compare(0, 4, "bam_") == 0 && kv.second.p) v.push_back(&kv.second);
    for (DBuf *d : v) { (void)hipFree(d->p); d->p = nullptr; d->bytes = 0; }
    if (ctx->twin_pool) { (void)hipFree(ctx->twin_pool); ctx->twin_pool = nullptr; ctx->twin_pool_cap = 0; }      // the device CIGAR array of a freed result
    return TELR_OK;
}
extern "C" int telr_device_mem(telr_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes)
{
    if (!ctx) return TELR_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    size_t fr = 0, tot = 0; HIPCHK(hipMemGetInfo(&fr, &tot));
    if (free_bytes) *free_bytes = (int64_t)fr;
    if (total_bytes) *total_bytes = (int64_t)tot;
    return TELR_OK;
}
// ---- one coordinate slice of a BAM (N > 1 ranks: every rank writes the slice of the job's file it holds the records of) ----
// The BGZF blocks of the slice as an image in the context's device memory (valid until the context's next writer call), the
// blocks' offsets inside it, and what the index needs of every record written: rank 0 merges those into the one .bai.
struct telr_bam_segment {
    telr_ctx *ctx = nullptr;
    const uint8_t *d_img = nullptr; uint64_t cbytes = 0, ubytes = 0;
    std::vector<uint64_t> coff;                       // [nblk + 1] offsets of the BGZF blocks inside the image
    std::vector<int32_t> tid, ts, te;                 // mapped records of the slice in file order
    std::vector<uint64_t> ustart;                     // [n + 1] their offsets in the slice's uncompressed stream (header included)
    int64_t n_unmapped = 0;
};
struct BamSliceOpt { const uint8_t *emit; int32_t with_header; telr_bam_segment *seg; };
static int bam_dev_impl(telr_ctx *ctx, const telr_result *r, const telr_seqset *queries, const telr_index *idx, const char *const *qnames,
                        const char *const *tnames, int32_t flags, const char *rg_id, const char *rg_sm, const char *rg_lb, const char *pg_line,
                        const char *bam_path, int32_t write_index, int32_t level, const BamSliceOpt *so = nullptr);
// one writer at a time per process, whole-file and slice calls alike: the call's timing / size records (g_bam_*), the NOMEM retry size and
// the code tables are process-wide
static std::mutex g_bam_mu;
extern "C" int telr_write_bam_dev(telr_ctx *ctx, const telr_result *r, const telr_seqset *queries, const telr_index *idx, const char *const *qnames,
                                  const char *const *tnames, int32_t flags, const char *rg_id, const char *rg_sm, const char *rg_lb, const char *pg_line,
                                  const char *bam_path, int32_t write_index, int32_t level)
{
    std::lock_guard<std::mutex> writer_lock(g_bam_mu);
    (void)hipGetLastError();          // a failed allocation of an EARLIER call leaves its error with the thread: not this call's
    int rc = bam_dev_impl(ctx, r, queries, idx, qnames, tnames, flags, rg_id, rg_sm, rg_lb, pg_line, bam_path, write_index, level);
    if (rc == TELR_E_NOMEM) {
        (void)hipGetLastError();
        mem_note(ctx, "telr_write_bam_dev: out of memory");
        ctx_release_map_scratch(ctx, g_bam_need);
        mem_note(ctx, "telr_write_bam_dev: after giving back scratch");
        rc = bam_dev_impl(ctx, r, queries, idx, qnames, tnames, flags, rg_id, rg_sm, rg_lb, pg_line, bam_path, write_index, level);
    }
    if (rc != TELR_OK && bam_path) {
        // no partial (or merely pre-sized) file at the output path: callers test for the file's existence (TELR_alignment.py:110-114)
        if (ctx) bam_sink_drop(ctx);
        (void)unlink(bam_path); (void)unlink((std::string(bam_path) + ".bai").c_str());
    }
    return rc;
}
static int bam_dev_impl(telr_ctx *ctx, const telr_result *r, const telr_seqset *queries, const telr_index *idx, const char *const *qnames,
                        const char *const *tnames, int32_t flags, const char *rg_id, const char *rg_sm, const char *rg_lb, const char *pg_line,
                        const char *bam_path, int32_t write_index, int32_t level, const BamSliceOpt *so)
{
    if (!ctx || !r || !queries || !idx || !idx->targets || !qnames || !tnames || (!bam_path && !so)) return TELR_E_ARG;
    if (level < 0) return TELR_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<float, std::milli>(now() - t0).count(); };
    const auto t_all = now();
    memset(&g_bam_times, 0, sizeof(g_bam_times));
    g_bam_need = (uint64_t)queries->total_bases * (level == 0 ? 6 : 9) + (uint64_t)r->ncig * 4 + (256u << 20);      // refined below once the stream's size is known
    result_wait(r);
    hipStream_t st = ctx->stream;
    const telr_seqset *tg = idx->targets;
    const int32_t nq = queries->n, nt = tg->n;
    const size_t n_mapped = r->alns.size();
    for (const telr_aln &a : r->alns) if (a.qid < 0 || a.qid >= nq || a.tid < 0 || a.tid >= nt) return TELR_E_ARG;
    // the CIGAR words (1.9 GB for a 30x set: 34 ms of PCIe) start their way back to the device before the host lays out records and names
    auto t0 = now();
    // ... unless the result kept them there (TELR_MF_KEEP_CIGARS): the mirrored array is used in place
    uint32_t *d_cig;
    static const bool no_twin = ab_on("bam_no_twin");
    const bool twin = r->d_cig && !r->twin_off && r->twin_n == r->ncig && !no_twin;
    g_bam_twin = twin ? 1 : 0;
    if (twin) { d_cig = r->d_cig; HIPCHK(hipDeviceSynchronize()); }      // its last pieces were copied on other streams
    else {
        TRY(ctx_buf_t(ctx, "bam_cig", r->ncig + 1, &d_cig));
        if (r->ncig) HIPCHK(hipMemcpyAsync(d_cig, r->cig, r->ncig * 4, hipMemcpyHostToDevice, st));
    }
    // ---- 1. records (+ pseudo records of the unmapped reads, in query order), names, header
    std::vector<telr_aln> recs(r->alns);
    size_t n_unmapped = 0;
    if (!(flags & TELR_SAM_NO_UNMAPPED)) {
        std::vector<uint8_t> has((size_t)nq, 0);
        for (const telr_aln &a : r->alns) has[a.qid] = 1;
        for (int q = 0; q < nq; ++q) if (!has[q]) { telr_aln u; memset(&u, 0, sizeof(u)); u.qid = q; u.tid = -1; u.qlen = queries->len[q]; recs.push_back(u); ++n_unmapped; }
    }
    const size_t nrec = recs.size();
    if (nrec >= (1u << 31)) return TELR_E_RANGE;
    std::vector<int64_t> qn_off((size_t)nq + 1); std::string qn_buf;
    { int64_t o = 0; for (int q = 0; q < nq; ++q) { qn_off[q] = o; size_t l = strlen(qnames[q]) + 1; if (l > 255) return TELR_E_ARG; o += (int64_t)l; } qn_off[nq] = o; qn_buf.resize((size_t)o);
      parallel_ranges(host_threads(), nq, [&](int, int a0, int a1) { for (int q = a0; q < a1; ++q) memcpy(&qn_buf[(size_t)qn_off[q]], qnames[q], (size_t)(qn_off[q + 1] - qn_off[q])); }); }
    std::vector<int32_t> tn_off((size_t)nt + 1); std::string tn_buf;
    { int32_t o = 0; for (int t = 0; t < nt; ++t) { tn_off[t] = o; o += (int32_t)strlen(tnames[t]) + 1; } tn_off[nt] = o; tn_buf.resize((size_t)o); for (int t = 0; t < nt; ++t) memcpy(&tn_buf[tn_off[t]], tnames[t], (size_t)(tn_off[t + 1] - tn_off[t])); }
    const std::string head = (so && !so->with_header) ? std::string() : bam_header(nt, tnames, tg->len.data(), rg_id, rg_sm, rg_lb, pg_line);
    const int rg_len = rg_id ? (int)strlen(rg_id) : 0;
    // ---- 2. upload
    telr_aln *d_alns; char *d_qn, *d_tn, *d_rg; int64_t *d_qnoff; int32_t *d_tnoff; BamInfo *d_info; uint32_t *d_size, *d_ord0, *d_ord; uint64_t *d_key, *d_key2, *d_szs, *d_ust, *d_rust;
    CrcTabs *d_tabs;
    TRY(ctx_buf_t(ctx, "bam_alns", nrec, &d_alns));
    TRY(ctx_buf_t(ctx, "bam_qn", qn_buf.size() + 1, &d_qn)); TRY(ctx_buf_t(ctx, "bam_tn", tn_buf.size() + 1, &d_tn)); TRY(ctx_buf_t(ctx, "bam_rg", (size_t)rg_len + 1, &d_rg));
    TRY(ctx_buf_t(ctx, "bam_qnoff", (size_t)nq + 1, &d_qnoff)); TRY(ctx_buf_t(ctx, "bam_tnoff", (size_t)nt + 1, &d_tnoff));
    TRY(ctx_buf_t(ctx, "bam_info", nrec, &d_info)); TRY(ctx_buf_t(ctx, "bam_size", nrec, &d_size)); TRY(ctx_buf_t(ctx, "bam_ord0", nrec, &d_ord0)); TRY(ctx_buf_t(ctx, "bam_ord", nrec, &d_ord));
    TRY(ctx_buf_t(ctx, "bam_key", nrec, &d_key)); TRY(ctx_buf_t(ctx, "bam_key2", nrec, &d_key2)); TRY(ctx_buf_t(ctx, "bam_szs", nrec + 1, &d_szs)); TRY(ctx_buf_t(ctx, "bam_ust", nrec + 1, &d_ust));
    TRY(ctx_buf_t(ctx, "bam_rust", nrec, &d_rust)); TRY(ctx_buf_t(ctx, "bam_tabs", 1, &d_tabs));
    { static CrcTabs T; static bool made = false; if (!made) { crc_tabs_make(T); made = true; } HIPCHK(hipMemcpyAsync(d_tabs, &T, sizeof(T), hipMemcpyHostToDevice, st)); }
    if (nrec) HIPCHK(hipMemcpyAsync(d_alns, recs.data(), nrec * sizeof(telr_aln), hipMemcpyHostToDevice, st));
    if (!qn_buf.empty()) HIPCHK(hipMemcpyAsync(d_qn, qn_buf.data(), qn_buf.size(), hipMemcpyHostToDevice, st));
    if (!tn_buf.empty()) HIPCHK(hipMemcpyAsync(d_tn, tn_buf.data(), tn_buf.size(), hipMemcpyHostToDevice, st));
    if (rg_len) HIPCHK(hipMemcpyAsync(d_rg, rg_id, (size_t)rg_len, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_qnoff, qn_off.data(), ((size_t)nq + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_tnoff, tn_off.data(), ((size_t)nt + 1) * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    g_bam_times.ms[0] = ms_since(t0); t0 = now();
    BamArgs A; memset(&A, 0, sizeof(A));
    A.alns = d_alns; A.nrec = (int32_t)nrec; A.n_mapped = (int32_t)n_mapped; A.cig = d_cig;
    A.q2 = queries->d_seq2; A.qn = queries->d_nmask; A.qboff = queries->d_boff; A.t2 = tg->d_seq2; A.tn = tg->d_nmask; A.tboff = tg->d_boff;
    A.qnames = d_qn; A.qname_off = d_qnoff; A.tnames = d_tn; A.tname_off = d_tnoff; A.rg = d_rg; A.rg_len = rg_len; A.flags = flags;
    A.info = d_info; A.rec_size = d_size; A.key = d_key; A.rec_ustart = d_rust; A.ubuf = nullptr;
    size_t n_ghost = 0;
    if (so && so->emit) {
        uint8_t *d_emit; TRY(ctx_buf_t(ctx, "bam_emit", nrec + 1, &d_emit));
        std::vector<uint8_t> h_emit(nrec, 1);             // (the pseudo records of the unmapped reads are always this rank's)
        for (size_t k = 0; k < n_mapped; ++k) { h_emit[k] = so->emit[k] ? 1 : 0; n_ghost += h_emit[k] ? 0 : 1; }
        if (nrec) HIPCHK(hipMemcpyAsync(d_emit, h_emit.data(), nrec, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        A.emit = d_emit;
    }
    uint64_t utotal = head.size();
    std::vector<uint32_t> h_order(nrec); std::vector<uint64_t> h_ustart(nrec + 1, head.size());
    if (nrec) {
        // ---- 3. sizes and keys
        if (n_mapped) hipLaunchKernelGGL(k_bam_scan, dim3((unsigned)n_mapped), dim3(64), 0, st, A);
        hipLaunchKernelGGL(k_bam_size, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, st, A);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
        g_bam_times.ms[1] = ms_since(t0); t0 = now();
        // ---- 4. coordinate sort (stable: ties keep the query order) and the offsets
        hipLaunchKernelGGL(k_iota_u32, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, st, d_ord0, (int32_t)nrec);
        size_t tb = 0;
        HIPCHK(rocprim::radix_sort_pairs(nullptr, tb, d_key, d_key2, d_ord0, d_ord, nrec, 0, 64, st));
        void *tmp; TRY(ctx_buf(ctx, "rp_tmp", tb, &tmp));
        HIPCHK(rocprim::radix_sort_pairs(tmp, tb, d_key, d_key2, d_ord0, d_ord, nrec, 0, 64, st));
        hipLaunchKernelGGL(k_bam_gather_sizes, dim3((unsigned)((nrec + 256) / 256)), dim3(256), 0, st, d_size, d_ord, (int32_t)nrec, d_szs);
        TRY((dev_exclusive_scan<uint64_t, uint64_t>(ctx, d_szs, d_ust, nrec + 1)));
        hipLaunchKernelGGL(k_bam_scatter_off, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, st, d_ust, d_ord, (int32_t)nrec, (uint64_t)head.size(), d_rust);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h_order.data(), d_ord, nrec * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(h_ustart.data(), d_ust, (nrec + 1) * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        for (size_t i = 0; i <= nrec; ++i) h_ustart[i] += head.size();
        utotal = h_ustart[nrec];
        g_bam_times.ms[2] = ms_since(t0); t0 = now();
    }
    // ---- 5. the uncompressed stream
    g_bam_need = level == 0 ? 2 * utotal + ((utotal / BAM_BLK + 1) * 31) : 3 * utotal + (utotal / BAM_BLK + 1) * 320;       // stream + slots + image
    uint8_t *d_u; TRY(ctx_buf_t(ctx, "bam_u", (size_t)utotal + 64, &d_u));
    HIPCHK(hipMemcpyAsync(d_u, head.data(), head.size(), hipMemcpyHostToDevice, st));
    A.ubuf = d_u; A.order = nullptr; A.s0 = 0;
    const size_t nblk = (size_t)((utotal + BAM_BLK - 1) / BAM_BLK);
    // The records are written in NG pieces of the sorted order (equal shares of the stream): the blocks a piece completes are
    // coded -- and on their way to the file -- while the next pieces are still being written (level >= 1; stored blocks: one go).
    const size_t NG = 16;
    std::vector<size_t> s_end(NG + 1, 0), b_done(NG + 1, 0);
    for (size_t g = 1; g <= NG; ++g) {
        s_end[g] = g == NG ? nrec : (size_t)(std::lower_bound(h_ustart.begin(), h_ustart.begin() + nrec, utotal / NG * g) - h_ustart.begin());
        if (s_end[g] < s_end[g - 1]) s_end[g] = s_end[g - 1];
        b_done[g] = g == NG ? nblk : std::min(nblk, (size_t)(h_ustart[s_end[g]] / BAM_BLK));        // every byte before record s_end[g] is final
    }
    // the pieces are written on a stream of their own, one behind the other (a latency-bound kernel that shares the device well
    // with the LDS-bound coder); the coder's stream waits for the piece a group needs
    hipStream_t sw = ctx->side[1];
    EvBag piece_bag, group_bag; piece_bag.v.assign(NG, nullptr);
    std::vector<hipEvent_t> &ev_piece = piece_bag.v;
    auto write_piece = [&](size_t g) {          // records [s_end[g], s_end[g + 1]) of the sorted order
        const size_t n = s_end[g + 1] - s_end[g];
        if (n) {
            BamArgs P = A; P.order = d_ord; P.s0 = (int32_t)s_end[g];
            hipLaunchKernelGGL(k_bam_write, dim3((unsigned)n), dim3(64), 0, sw, P);
        }
        if (hipEventCreateWithFlags(&ev_piece[g], hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(ev_piece[g], sw);
    };
    HIPCHK(hipMemsetAsync(d_u + utotal, 0, 64, st));
    if (level == 0) { if (nrec) hipLaunchKernelGGL(k_bam_write, dim3((unsigned)nrec), dim3(64), 0, st, A); }
    else {
        HIPCHK(hipEventRecord(ctx->ev_fork, st)); HIPCHK(hipStreamWaitEvent(sw, ctx->ev_fork, 0));       // header, offsets and sort order are in place
        for (size_t g = 0; g < NG; ++g) write_piece(g);
        if (ev_piece[0]) HIPCHK(hipStreamWaitEvent(st, ev_piece[0], 0));
    }
    HIPCHK(hipGetLastError());
    if (ctx->debug) { HIPCHK(hipStreamSynchronize(st)); }
    g_bam_times.ms[3] = ms_since(t0); t0 = now();
    // ---- 6. BGZF blocks
    uint64_t cbytes = 0;
    uint8_t *d_c = nullptr;
    std::vector<uint64_t> coff(nblk + 1);
    StreamProgress *prog = nullptr; std::thread producer;
    if (level == 0) {
        cbytes = utotal + (uint64_t)nblk * 31;
        TRY(ctx_buf_t(ctx, "bam_c", (size_t)cbytes + 64, &d_c));
        if (nblk) hipLaunchKernelGGL(k_bgzf_store, dim3((unsigned)nblk), dim3(256), 0, st, d_u, utotal, d_tabs, d_c);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
        for (size_t b = 0; b < nblk; ++b) coff[b] = (uint64_t)b * (BAM_BLK + 31);
        coff[nblk] = cbytes;
    } else {
        int32_t *d_rec0; DeflTabs *d_T; uint32_t *d_hist, *d_csize; uint64_t *d_cs64, *d_coff; uint8_t *d_slots;
        TRY(ctx_buf_t(ctx, "bam_rec0", nblk, &d_rec0)); TRY(ctx_buf_t(ctx, "bam_T", 1, &d_T)); TRY(ctx_buf_t(ctx, "bam_hist", (size_t)DEFL_NCLS * 288, &d_hist));
        TRY(ctx_buf_t(ctx, "bam_csize", nblk, &d_csize)); TRY(ctx_buf_t(ctx, "bam_cs64", nblk + 1, &d_cs64)); TRY(ctx_buf_t(ctx, "bam_coff", nblk + 1, &d_coff));
        TRY(ctx_buf_t(ctx, "bam_slots", nblk * (size_t)DEFL_SLOT, &d_slots));
        hipLaunchKernelGGL(k_blk_first_rec, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, st, d_ust, (int32_t)nrec, (uint64_t)head.size(), (int32_t)nblk, d_rec0);
        static DeflTabs T;                   // 8 KB: not on the stack of a ctypes caller's thread
        memset(&T, 0, sizeof(T)); defl_len_syms(T.len_sym);
        HIPCHK(hipMemcpyAsync(d_T, &T, sizeof(T), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)DEFL_NCLS * 288 * 4, st));
        // symbol statistics per field class from every `stride`-th block (at most ~4096 blocks: 270 MB of a 30x read set)
        // ... of the FIRST piece (the records are sorted by position: the pieces are alike), so that coding can start behind it
        const size_t nb_first = b_done[1] ? b_done[1] : nblk;
        if (!b_done[1]) {       // a stream of less than one block per piece: all of it first
            for (size_t g = 1; g < NG; ++g) { if (ev_piece[g]) HIPCHK(hipStreamWaitEvent(st, ev_piece[g], 0)); b_done[g] = nblk; }
        }
        const int32_t stride = (int32_t)std::max<size_t>(1, nb_first / 4096);
        const unsigned nsamp = (unsigned)((nb_first + stride - 1) / stride);
        hipLaunchKernelGGL(k_bam_hist, dim3(nsamp), dim3(DEFL_THREADS), 0, st, d_u, utotal, (uint64_t)head.size(), d_ust, (int32_t)nrec, d_rec0, stride, d_T, d_hist);
        HIPCHK(hipGetLastError());
        std::vector<uint32_t> h_hist((size_t)DEFL_NCLS * 288);
        HIPCHK(hipMemcpyAsync(h_hist.data(), d_hist, h_hist.size() * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        TRY(defl_tables_from_hist(h_hist.data(), T));
        HIPCHK(hipMemcpyAsync(d_T, &T, sizeof(T), hipMemcpyHostToDevice, st));
        // The blocks are coded in groups; a group's sizes come home, the host lays the group out behind the previous one and a
        // second stream moves it into the file image, whose finished prefix streams out to the file while the next groups are
        // still being coded (7. below).  The image buffer is sized for the worst case (every block stored).
        TRY(ctx_buf_t(ctx, "bam_c", (size_t)utotal + nblk * 31 + 64, &d_c));
        uint32_t *h_csize; uint64_t *h_coff;
        TRY(ctx_hbuf_t(ctx, "bam_hcsize", nblk + 1, &h_csize)); TRY(ctx_hbuf_t(ctx, "bam_hcoff", nblk + 1, &h_coff));
        std::vector<hipEvent_t> &evg = group_bag.v; std::vector<size_t> gb0, gnb;          // group g: blocks [gb0, gb0 + gnb), complete once piece g is written
        for (size_t g = 0; g < NG; ++g) {
            if (g && ev_piece[g]) HIPCHK(hipStreamWaitEvent(st, ev_piece[g], 0));
            const size_t b0 = b_done[g], nb = b_done[g + 1] - b_done[g];
            if (!nb) continue;
            hipLaunchKernelGGL(k_bgzf_deflate, dim3((unsigned)nb), dim3(DEFL_THREADS), 0, st, d_u, utotal, (uint64_t)head.size(), d_ust, (int32_t)nrec, d_rec0, d_T, d_tabs, d_slots, d_csize, (uint32_t)b0);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(h_csize + b0, d_csize + b0, nb * 4, hipMemcpyDeviceToHost, st));
            hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); HIPCHK(hipEventRecord(e, st)); evg.push_back(e);
            gb0.push_back(b0); gnb.push_back(nb);
        }
        prog = new StreamProgress();
        StreamProgress *pg = prog; uint64_t *coff_p = coff.data(); hipStream_t st2 = ctx->side[0]; const int device = ctx->device;
        BamSink *sk = bam_path && ctx->bam_sink && ctx->bam_sink->path == bam_path ? ctx->bam_sink : nullptr;
        const std::vector<hipEvent_t> evg_copy = evg;          // (the bag destroys them after the producer has been joined)
        producer = std::thread([=]() mutable {
            (void)hipSetDevice(device);
            uint64_t off = 0; bool ok = true;
            for (size_t g = 0; g < gb0.size() && ok; ++g) {
                const size_t b0 = gb0[g], nb = gnb[g];
                if (hipEventSynchronize(evg_copy[g]) != hipSuccess) { ok = false; break; }
                for (size_t b = b0; b < b0 + nb; ++b) { h_coff[b] = off; coff_p[b] = off; off += h_csize[b]; }
                // the file's length, extrapolated from the blocks coded so far (records are sorted by position: the groups are alike; 1.5 % + 4 MB on top)
                if (sk) sk->set_limit(b0 + nb >= nblk ? (size_t)off + 28 : (size_t)((double)off * (double)nblk / (double)(b0 + nb) * 1.015) + ((size_t)4 << 20));
                if (hipMemcpyAsync(d_coff + b0, h_coff + b0, nb * 8, hipMemcpyHostToDevice, st2) != hipSuccess) { ok = false; break; }
                hipLaunchKernelGGL(k_bgzf_compact, dim3((unsigned)nb), dim3(256), 0, st2, d_slots, d_csize, d_coff, d_c, (uint32_t)b0);
                if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st2) != hipSuccess) { ok = false; break; }
                progress_set(pg, off, 0, 0);
            }
            coff_p[nblk] = off;
            progress_set(pg, off, ok ? 1 : -1, off);
        });
        (void)d_cs64;
    }
    g_bam_times.ms[4] = ms_since(t0); t0 = now();
    // ---- 7. the file image streams out (behind the groups still being coded); the index is built by a host thread meanwhile
    std::string bai; std::thread bai_th;
    float bai_ms = 0;
    // the index is laid out with stream offsets while the blocks are still being coded; their file offsets go in at the end
    std::vector<size_t> bai_fix;
    if (write_index && !so) bai_th = std::thread([&] { auto tb0 = now(); bai_build(recs, h_order.data(), nrec, n_unmapped, h_ustart.data(), nt, tg->len.data(), bai, bai_fix); bai_ms = ms_since(tb0); });
    std::thread bai_starter;
    if (prog) bai_starter = std::thread([&] { if (producer.joinable()) producer.join(); });        // the block offsets are complete when the producer is
    static const uint8_t eof_blk[28] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    if (so) {
        // slice mode: the image stays on the device (telr_bam_segment_write puts it at its place in the job's file once the
        // sizes of the slices before it are known); the records' coordinates and stream offsets go with it for the one index
        if (bai_starter.joinable()) bai_starter.join();
        if (producer.joinable()) producer.join();
        int rc = TELR_OK;
        if (prog) { if (prog->state < 0) rc = TELR_E_HIP; cbytes = prog->total; delete prog; }
        HIPCHK(hipDeviceSynchronize());
        telr_bam_segment *S = so->seg;
        S->ctx = ctx; S->d_img = d_c; S->cbytes = cbytes; S->ubytes = utotal; S->coff = coff; S->n_unmapped = (int64_t)n_unmapped;
        const size_t nw = nrec - n_ghost - n_unmapped;                       // mapped records of the slice, first in the sorted order
        S->tid.resize(nw); S->ts.resize(nw); S->te.resize(nw); S->ustart.assign(h_ustart.begin(), h_ustart.begin() + nw + 1);
        for (size_t i = 0; i < nw; ++i) { const telr_aln &a = recs[h_order[i]]; S->tid[i] = a.tid; S->ts[i] = a.ts; S->te[i] = a.te > a.ts ? a.te : a.ts + 1; }
        g_bam_times.ms[7] = ms_since(t_all);
        return rc;
    }
    int rc = sink_to_file(ctx, d_c, cbytes, prog, eof_blk, 28, bam_path);
    g_bam_times.ms[5] = ms_since(t0);
    if (bai_starter.joinable()) bai_starter.join();
    if (bai_th.joinable()) bai_th.join();
    if (write_index) bai_finish(bai, bai_fix, coff.data(), nblk);
    bam_sink_drop(ctx);
    g_bam_times.ms[6] = bai_ms;
    if (prog) { if (prog->state < 0 && rc == TELR_OK) rc = TELR_E_HIP; delete prog; }
    if (rc == TELR_OK && write_index) {
        const std::string bai_path = std::string(bam_path) + ".bai";
        FILE *f = fopen(bai_path.c_str(), "wb");
        if (!f) rc = TELR_E_ARG; else { fwrite(bai.data(), 1, bai.size(), f); fclose(f); }
    }
    g_bam_times.ms[7] = ms_since(t_all);
    return rc;
}

// ---- the slice entry points (include/telr_hip.h) ---------------------------------------------------------------------
extern "C" int telr_write_bam_slice(telr_ctx *ctx, const telr_result *r, const telr_seqset *queries, const telr_index *idx, const char *const *qnames,
                                    const char *const *tnames, int32_t flags, const char *rg_id, const char *rg_sm, const char *rg_lb, const char *pg_line,
                                    const uint8_t *emit, int32_t with_header, int32_t level, telr_bam_segment **out)
{
    if (!out || level < 1) return TELR_E_ARG;
    std::lock_guard<std::mutex> lk(g_bam_mu);               // (the same lock as telr_write_bam_dev: the writer's records are process-wide)
    (void)hipGetLastError();
    telr_bam_segment *S = new telr_bam_segment();
    BamSliceOpt so{ emit, with_header, S };
    int rc = bam_dev_impl(ctx, r, queries, idx, qnames, tnames, flags, rg_id, rg_sm, rg_lb, pg_line, nullptr, 0, level, &so);
    if (rc == TELR_E_NOMEM) {
        (void)hipGetLastError();
        ctx_release_map_scratch(ctx, g_bam_need);
        rc = bam_dev_impl(ctx, r, queries, idx, qnames, tnames, flags, rg_id, rg_sm, rg_lb, pg_line, nullptr, 0, level, &so);
    }
    if (rc != TELR_OK) { delete S; return rc; }
    *out = S;
    return TELR_OK;
}
extern "C" void telr_bam_segment_free(telr_bam_segment *s) { delete s; }
// out[0] bytes of the slice in the file, [1] mapped records, [2] unmapped reads, [3] bytes of its uncompressed stream
extern "C" int telr_bam_segment_info(const telr_bam_segment *s, int64_t *out)
{
    if (!s || !out) return TELR_E_ARG;
    out[0] = (int64_t)s->cbytes; out[1] = (int64_t)s->tid.size(); out[2] = s->n_unmapped; out[3] = (int64_t)s->ubytes;
    return TELR_OK;
}
static uint64_t segment_voff(const telr_bam_segment *s, uint64_t file_off, uint64_t u)
{
    const size_t nblk = s->coff.size() - 1, b = (size_t)(u / BAM_BLK);
    if (b >= nblk) return (file_off + s->cbytes) << 16;
    return (file_off + s->coff[b]) << 16 | (u - (uint64_t)b * BAM_BLK);
}
// what the job's index needs of the slice, with the slice at `file_off` of the file: per mapped record (file order) the
// reference, start, end and the virtual offset of its first byte; v_end = the virtual offset behind the last mapped record
extern "C" int telr_bam_segment_entries(const telr_bam_segment *s, int64_t file_off, int32_t *tid, int32_t *ts, int32_t *te, uint64_t *vb, uint64_t *v_end)
{
    if (!s || file_off < 0 || !v_end || s->coff.empty()) return TELR_E_ARG;
    const size_t n = s->tid.size();
    if (n && (!tid || !ts || !te || !vb)) return TELR_E_ARG;
    for (size_t i = 0; i < n; ++i) { tid[i] = s->tid[i]; ts[i] = s->ts[i]; te[i] = s->te[i]; vb[i] = segment_voff(s, (uint64_t)file_off, s->ustart[i]); }
    *v_end = segment_voff(s, (uint64_t)file_off, s->ustart[n]);
    return TELR_OK;
}
// the slice's image -> bytes [file_off, file_off + size) of `path` (an existing file: rank 0 creates it, every rank writes its
// own range); is_last: the BGZF end-of-file block behind it.  The range is allocated first (ENOSPC is an error code here, never
// a SIGBUS), mapped and pre-faulted, and filled through the pinned ring by the host pool.
extern "C" int telr_bam_segment_write(telr_ctx *ctx, const telr_bam_segment *s, const char *path, int64_t file_off, int32_t is_last)
{
    if (!ctx || !s || !path || file_off < 0 || s->ctx != ctx) return TELR_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    static const uint8_t eof_blk[28] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const size_t tail = is_last ? 28 : 0, len = (size_t)s->cbytes + tail;
    int fd = open(path, O_RDWR);
    if (fd < 0) { ctx->err = std::string("cannot open ") + path; return TELR_E_ARG; }
    int rc = TELR_OK;
    uint8_t *map = nullptr; size_t map_len = 0; const uint64_t al = (uint64_t)file_off & ~(uint64_t)4095;
    if (len) {
        if (posix_fallocate(fd, (off_t)file_off, (off_t)len) != 0) { ctx->err = std::string("no space for ") + path; close(fd); return TELR_E_ARG; }
        map_len = (size_t)((uint64_t)file_off + len - al);
        void *m = mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)al);
        if (m != MAP_FAILED) { map = (uint8_t*)m; (void)madvise(map, map_len, 23 /* MADV_POPULATE_WRITE */); }
    }
    uint64_t total = 0;
    rc = stream_to_file(ctx, s->d_img, s->cbytes, nullptr, eof_blk, tail, path, fd, map ? map + ((uint64_t)file_off - al) : nullptr, map ? len : 0, &total, nullptr, nullptr,
                        (uint64_t)file_off, true);
    if (map) munmap(map, map_len);
    close(fd);
    if (rc == TELR_OK && total != s->cbytes) rc = TELR_E_HIP;
    return rc;
}
// The .bai of a file whose records are described by arrays in FILE order (n mapped records: reference, start, end, virtual
// offset of the first byte; v_end: the virtual offset behind the last of them; n_unmapped reads follow): what rank 0 writes
// from the entries of all slices.  Same bins, chunk merging, linear index and metadata pseudo-bin as the one-rank writer.
extern "C" int telr_bai_write(const char *bai_path, int64_t n, const int32_t *tid, const int32_t *ts, const int32_t *te, const uint64_t *vb, uint64_t v_end,
                              int64_t n_unmapped, int32_t n_targets, const int32_t *t_len)
{
    if (!bai_path || n < 0 || n_targets < 0 || (n && (!tid || !ts || !te || !vb)) || (n_targets && !t_len)) return TELR_E_ARG;
    std::string bai;
    auto put32 = [&](uint32_t v) { bai.append((const char*)&v, 4); };
    auto put64 = [&](std::string &d, uint64_t v) { d.append((const char*)&v, 8); };
    bai = "BAI\1"; put32((uint32_t)n_targets);
    int64_t i = 0;
    struct Ch { uint32_t bin; uint64_t vb, ve; };
    std::vector<Ch> chs;
    for (int t = 0; t < n_targets; ++t) {
        chs.clear();
        const int n_lin = (t_len[t] >> 14) + 1;
        std::vector<uint64_t> lin(n_lin, 0);
        int max_lin = 0; uint64_t ref_beg = 0, ref_end = 0, n_map = 0; bool any = false;
        while (i < n && tid[i] == t) {
            const uint64_t b = vb[i], e_ = i + 1 < n ? vb[i + 1] : v_end;
            const int e = te[i] > ts[i] ? te[i] : ts[i] + 1;
            chs.push_back(Ch{ (uint32_t)reg2bin(ts[i], e), b, e_ });
            const int w0 = ts[i] >> 14, w1 = (e - 1) >> 14;
            for (int wv = w0; wv <= w1 && wv < n_lin; ++wv) { if (lin[wv] == 0 || b < lin[wv]) lin[wv] = b; if (wv + 1 > max_lin) max_lin = wv + 1; }
            if (!any) { ref_beg = b; any = true; }
            ref_end = e_; ++n_map; ++i;
        }
        if (i < n && tid[i] < t) return TELR_E_ARG;                 // not in file order
        std::stable_sort(chs.begin(), chs.end(), [](const Ch &x, const Ch &y) { return x.bin < y.bin; });
        std::string body; uint32_t nbin = 0;
        for (size_t c0 = 0; c0 < chs.size(); ) {
            size_t c1 = c0; std::vector<std::pair<uint64_t, uint64_t>> ch;
            while (c1 < chs.size() && chs[c1].bin == chs[c0].bin) {
                if (!ch.empty() && (ch.back().second >> 16) == (chs[c1].vb >> 16)) ch.back().second = chs[c1].ve; else ch.push_back(std::make_pair(chs[c1].vb, chs[c1].ve));
                ++c1;
            }
            uint32_t bin = chs[c0].bin, nc = (uint32_t)ch.size();
            body.append((const char*)&bin, 4); body.append((const char*)&nc, 4);
            for (auto &c : ch) { put64(body, c.first); put64(body, c.second); }
            ++nbin; c0 = c1;
        }
        put32(nbin + (any ? 1 : 0));
        bai += body;
        if (any) { put32(37450u); put32(2u); put64(bai, ref_beg); put64(bai, ref_end); put64(bai, n_map); put64(bai, 0); }
        for (int wv = 1; wv < max_lin; ++wv) if (lin[wv] == 0) lin[wv] = lin[wv - 1];
        put32((uint32_t)max_lin);
        for (int wv = 0; wv < max_lin; ++wv) put64(bai, lin[wv]);
    }
    if (i != n) return TELR_E_ARG;
    put64(bai, (uint64_t)n_unmapped);
    FILE *f = fopen(bai_path, "wb");
    if (!f) return TELR_E_ARG;
    const bool ok = fwrite(bai.data(), 1, bai.size(), f) == bai.size();
    return (fclose(f) == 0 && ok) ? TELR_OK : TELR_E_ARG;
}
