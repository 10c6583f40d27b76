/*
 * telr_hip.h — C ABI of libtelrhip.so, the MI355X (gfx950) alignment engine that
 * replaces the seven aligner subprocess call sites of bergmanlab/TELR.
 *
 * The reference has no FFI: its boundary is `subprocess` argv + a SAM/PAF text
 * file.  Each entry point below cites the reference call site(s) it stands in
 * for (paths relative to the reference checkout):
 *
 *   S1  ngmlr -r R -q Q -x {ont,pacbio} ...            src/telr/TELR_alignment.py:31-51
 *   S2  minimap2 --cs --MD -Y -L -ax P R Q             src/telr/TELR_alignment.py:69-82
 *   S3  minimap2 -t N -ax P -r2k CNS READS             src/telr/TELR_assembly.py:199-212
 *   S4  minimap2 -cx P --secondary=no -v 0 SUBJ QRY    src/telr/TELR_te.py:68-78
 *   S5  minimap2 -cx P CONTIG LIB -v 0 -t T            src/telr/TELR_te.py:119-132
 *   S6  minimap2 -a -x P -v 0 SUBJ QRY                 src/telr/TELR_te.py:504-506
 *   S7  minimap2 -cx asm10 -v 0 -N 10 REF FLANK        src/telr/TELR_liftover.py:253-266
 *   D   samtools depth -aa -r chr:S-E  -> median       src/telr/TELR_te.py:870-884
 *
 * Conventions: every function returns 0 on success and a negative TELR_E_* code
 * on failure (telr_strerror() gives the text).  An empty result is NOT an
 * error (the reference treats an empty PAF as "locus not passed",
 * TELR_te.py:79-81).  The caller owns every input buffer (host pointers, may be
 * freed after the call returns); the library owns outputs until the matching
 * telr_*_free().  Device buffers never escape.  No torch types, no C++ types.
 */
#ifndef TELR_HIP_H
#define TELR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TELR_OK            0
#define TELR_E_NODEVICE   -1   /* no HIP device / HIP runtime error at init  */
#define TELR_E_HIP        -2   /* HIP runtime error (see telr_last_error)    */
#define TELR_E_ARG        -3   /* invalid argument                           */
#define TELR_E_RANGE      -4   /* input exceeds the engine's coordinate bits */
#define TELR_E_NOMEM      -5
#define TELR_E_IO         -6   /* a file could not be opened / mapped / written (telr_fasta_load: errno is the caller's to read) */

/* ---- alignment record flags -------------------------------------------- */
#define TELR_F_PRIMARY     0x1   /* parent == self                                  */
#define TELR_F_SECONDARY   0x2   /* overlaps a better chain on the query (SAM 0x100) */
#define TELR_F_SUPPL       0x4   /* primary chain that is not the best (SAM 0x800)   */
#define TELR_F_REV         0x8   /* query aligned on the reverse strand              */

/* ---- index options (minimap2 -k/-w/-H) ---------------------------------- */
typedef struct telr_idx_opt {
    int32_t k;            /* k-mer length, 4..28                                  */
    int32_t w;            /* minimizer window, 1..255                             */
    int32_t is_hpc;       /* homopolymer-compressed k-mers (map-pb)               */
    int32_t bucket_bits;  /* reserved (ignored): seeding probes an open-addressing table */
} telr_idx_opt;

/* ---- mapping options (one struct for all seven call sites) --------------- */
typedef struct telr_map_opt {
    /* seeding */
    float   mid_occ_frac;     /* -f   : fraction of most frequent minimizers ignored  */
    int32_t min_mid_occ;      /* -U lo                                                 */
    int32_t max_mid_occ;      /* -U hi                                                 */
    /* chaining */
    int32_t max_gap;          /* -g   : max reference/query distance between anchors   */
    int32_t bw;               /* -r   : max diagonal difference between anchors        */
    int32_t chain_lookback;   /* predecessors examined per anchor (multiple of 64)     */
    int32_t min_cnt;          /* -n   : min anchors per chain                          */
    int32_t min_chain_score;  /* -m                                                    */
    int32_t chain_gap_q8;     /* gap penalty per diagonal-difference base, Q8 fixed    */
    int32_t chain_skip_q8;    /* penalty per skipped base, Q8 fixed                    */
    /* chain selection */
    float   mask_level;       /* -M                                                    */
    float   pri_ratio;        /* -p                                                    */
    int32_t best_n;           /* -N                                                    */
    int32_t secondary;        /* --secondary=yes|no                                    */
    /* base-level alignment */
    int32_t a, b;             /* -A -B                                                 */
    int32_t q, e, q2, e2;     /* -O q,q2  -E e,e2                                      */
    int32_t sc_ambi;          /* penalty against an ambiguous base                     */
    int32_t zdrop;            /* -z                                                    */
    int32_t min_dp_max;       /* -s                                                    */
    int32_t min_ksw_len;      /* min gap-fill segment length                           */
    int32_t ext_max;          /* max bases an end extension may consume on the query   */
    int32_t ext_band;         /* half band width of end extensions (<= 31 with bw_long; presets: 31, ngmlr-ont 63) */
    int32_t flags;            /* TELR_MF_*                                             */
    int32_t fill_band_q4;     /* first-pass half band of a gap fill: 2 + q4*floor(sqrt(min(m,n)))/16;
                                 0 = 8.  Paths that come close to a band edge are re-aligned with the wide band. */
    int32_t fill_margin;      /* "close" = within this many diagonals of a band edge (0 = touching it)        */
    /* sub-read voting (NGMLR's candidate search, Sedlazeck 2018: the read is cut into sub-reads, the k-mer hits of a sub-read
     * vote for reference regions, only the best-voted regions are kept).  Applies to all-vs-all calls (qtarget == NULL, no
     * TELR_MF_PER_TARGET); 0 = off.  The hits of the minimizers whose last base lies in query bases [s*vote_len, (s+1)*vote_len)
     * vote into diagonal bins of 2^vote_bin_shift bases (per strand; 1024 bins, folded); a hit stays an anchor iff its bin and
     * the two next to it hold >= max(vote_min, ceil(vote_frac_q8 / 256 * votes of the sub-read's fullest bin)) hits. */
    int32_t vote_len;
    int32_t vote_bin_shift;
    int32_t vote_min;
    int32_t vote_frac_q8;
    /* long join (minimap2 -r500,20000: the second number): anchors are chained within max(bw, bw_long) diagonals, so that a
     * read across a multi-kb insertion or deletion is ONE chain; `bw` still bounds the gap fills.  A fill whose two lengths
     * differ by more than `bw` is aligned as two banded halves joined by one long gap at its best position.
     * 0 = off (asm10, ngmlr-*: NGMLR splits reads at SV breakpoints). */
    int32_t bw_long;
    /* convex gap cost (NGMLR, Sedlazeck 2018 Methods; `ngmlr -x ont`, TELR_alignment.py:28-51): cx_scale > 0 replaces the
     * two-piece affine cost (q, e, q2, e2 are then unused by the base-level DP) by: a gap pays cx_open once, and the gap base that
     * makes a gap of length i one longer pays ext(i) = max(cx_ext_min, cx_ext_max - cx_decay * i) -- NGMLR keeps the current
     * length with every gap cell, so this is the affine recurrence with a length-dependent extension.  All four numbers, and
     * every score inside the DP (a, b, sc_ambi, zdrop times cx_scale), are in 1/cx_scale of the unit of a / b / dp_score; a
     * record's dp_score is the sum of its segments' scores divided by cx_scale, rounded half up.  ngmlr-ont (NGMLR: match 1,
     * mismatch 1, open 1, extension 1 -> 0.5, decay 0.15; the preset's unit is half of NGMLR's): cx_scale 10, open 20,
     * ext 20 -> 10, decay 3.  ngmlr-pacbio (match 2, mismatch 5, open 5, extension 5 -> 1): cx_scale 20, open 100, ext 100 -> 20,
     * decay 3.  cx_scale = 0 over either preset gives the two-piece envelope q / e / q2 / e2 of round 3. */
    int32_t cx_scale, cx_open, cx_ext_max, cx_ext_min, cx_decay;
} telr_map_opt;

#define TELR_MF_CIGAR      0x1   /* -c / -a : run base-level alignment               */
#define TELR_MF_PER_TARGET 0x2   /* select/rank chains separately for every target:
                                    one call answers "QRY vs each of N subjects"
                                    (S5: the TE library against every contig)        */
#define TELR_MF_FAITHFUL   0x4   /* CPU oracle only (telr_map rejects it): drop the speed-motivated bounds of the spec --
                                    gap fills over the whole -r band, end extensions over the whole remaining read in a
                                    band of -r diagonals, chaining look-back 5000.  The reference point against which
                                    the tuned presets are gated (tests/test_faithful_gate.py).                          */

#define TELR_MF_KEEP_CIGARS 0x8  /* engine only (the oracle ignores it): the result keeps its CIGAR array on the device as well
                                    (same offsets; ~1.2 bytes per query base reserved), so that telr_write_bam_dev on the same
                                    context need not upload it again.  Records and CIGARs on the host are what they are without it. */

/* ---- one alignment (PAF line / SAM record worth of numbers), 88 bytes ---- */
typedef struct telr_aln {
    int32_t  qid;        /* query index in the query set                          */
    int32_t  tid;        /* target index in the index                             */
    int32_t  qlen;
    int32_t  qs, qe;     /* query interval on the FORWARD query strand, [qs,qe)   */
    int32_t  tlen;
    int32_t  ts, te;     /* target interval, [ts,te)                              */
    int32_t  mlen;       /* matching bases (PAF col 10)                           */
    int32_t  blen;       /* alignment block length (PAF col 11)                   */
    int32_t  score;      /* chaining score   (s1)                                 */
    int32_t  subsc;      /* best secondary chaining score (s2)                    */
    int32_t  dp_score;   /* DP alignment score (AS)                               */
    int32_t  cnt;        /* anchors on the chain (cm)                             */
    int32_t  n_sub;      /* number of suboptimal chains                           */
    int32_t  parent;     /* index (within this query's records) of the parent     */
    int32_t  n_cigar;    /* number of CIGAR ops                                   */
    int32_t  flags;      /* TELR_F_*                                              */
    int64_t  cigar_off;  /* offset of the first op in the result's cigar array    */
    int32_t  mapq;
    int32_t  n_ambi;     /* reserved (0): ambiguous columns count as mismatches   */
} telr_aln;

/* CIGAR op encoding: len<<4 | op, op 0=M 1=I 2=D (BAM numbering) */

typedef struct telr_ctx    telr_ctx;     /* one per process per device            */
typedef struct telr_seqset telr_seqset;  /* 2-bit packed sequences resident in HBM */
typedef struct telr_index  telr_index;   /* minimizer index resident in HBM        */
typedef struct telr_result telr_result;  /* host-side alignment records + CIGARs   */

/* ---- context ------------------------------------------------------------- */
int  telr_init(int device, telr_ctx **out);
/* A second context for a host thread of its own: contexts are not re-entrant, different contexts are independent, sequence sets
 * and indexes may be used from either.  Its streams have the device's lowest priority: the kernels of the process' other
 * context are dispatched first (the loci leg runs its one large realignment, S6, behind the small S4 / S5 / S7 calls). */
int  telr_init_background(int device, telr_ctx **out);
/* Scratch is grow-only per context (a 30x read set leaves 150-250 GB behind): give all of it back -- the next call sizes it
 * again, ~2 ms per GB -- and the device's free / total memory, for a process whose contexts share one device. */
int  telr_release_scratch(telr_ctx *ctx);
int  telr_device_mem(telr_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes);
void telr_destroy(telr_ctx *ctx);
const char *telr_strerror(int code);
const char *telr_last_error(const telr_ctx *ctx);   /* text of the last HIP error   */
int  telr_device_name(const telr_ctx *ctx, char *buf, int buflen);

/* ---- presets: the -x values the reference passes -------------------------- */
/* name in {"map-ont","map-pb","asm10","ngmlr-ont","ngmlr-pacbio"}; returns TELR_E_ARG otherwise.
 * ngmlr-*: the stage-1 default of the reference (`ngmlr -x ont|pacbio`, TELR_alignment.py:28-51): (w,k) = (5,13)
 * minimizers (NGMLR's 13-mers at every third position), sub-read voting, NGMLR's convex gap cost in exact form (cx_*). */
int  telr_preset(const char *name, telr_idx_opt *io, telr_map_opt *mo);

/* ---- sequence sets --------------------------------------------------------- */
/* n sequences given as one concatenated ASCII buffer; seq i = ascii[off[i] .. off[i]+len[i]).
 * Packs to 2 bits/base + an ambiguity bitmask (host threads, 32 bases per AVX2 step where available; pinned grow-only
 * staging in the context) and uploads to HBM, chunk by chunk while the next chunks are packed. */
int  telr_seqset_create(telr_ctx *ctx, int32_t n, const char *ascii,
                        const int64_t *off, const int32_t *len, telr_seqset **out);
void telr_seqset_free(telr_seqset *s);
/* a new set holding copies of sequences idx[0..n) of `parent` (repeats allowed), made on the device from the packed
 * form: no host packing, no upload.  Replaces `seqtk subseq` of window reads out of the read file
 * (TELR_assembly.py:419-456) when the reads are already resident for stage 1. */
int  telr_seqset_subset(telr_ctx *ctx, const telr_seqset *parent, int32_t n, const int32_t *idx, telr_seqset **out);
/* the same with an orientation per copy: sequence i of the new set is the REVERSE COMPLEMENT of parent[idx[i]] when rc[i] != 0
 * (rc NULL = telr_seqset_subset).  Replaces the reverse-complemented contig FASTA of the per-locus realignment
 * (`realignment()` maps the window reads to the contig and to its reverse complement, TELR_te.py:495-515, 644-646): the
 * contigs are turned on the device, from the packed form they already have there. */
int  telr_seqset_subset_rc(telr_ctx *ctx, const telr_seqset *parent, int32_t n, const int32_t *idx, const uint8_t *rc, telr_seqset **out);
int64_t telr_seqset_bases(const telr_seqset *s);
int32_t telr_seqset_count(const telr_seqset *s);
/* The packed form itself, for moving sequences between the GPUs of a node without ever unpacking them (the N > 1 hand-offs of
 * window reads to the owner of their locus -- TELR_assembly.py:384-462 does this through the shared file system --
 * and the reads of a coordinate slice to the rank that writes that slice of the BAM, TELR_alignment.py:103-114).  The layout is
 * a pure function of the lengths: sequence i starts at base offset B(i) = sum over j < i of ceil(len[j] / 64) * 64; 2-bit
 * codes, 16 bases per 32-bit word (base b of the set in bits 2(b % 16).. of word b / 16), ambiguity mask 32 bases per word.
 * Two sets placed end to end are therefore the set of the concatenated lengths.
 * telr_seqset_packed: DEVICE pointers to the two word arrays of `s` (valid until it is freed) and their lengths in words
 * (padded_bases / 16 and padded_bases / 32).  telr_seqset_from_packed: a new set of n sequences with the given lengths (host
 * array) whose word arrays are copied device-to-device from d_seq2 / d_nmask (device pointers, e.g. the receive buffer of an
 * all-to-all; nwords2 / nwordsn must equal the layout's sizes). */
int  telr_seqset_packed(const telr_seqset *s, const void **d_seq2, const void **d_nmask, int64_t *nwords2, int64_t *nwordsn);
int  telr_seqset_from_packed(telr_ctx *ctx, int32_t n, const int32_t *len, const void *d_seq2, int64_t nwords2,
                             const void *d_nmask, int64_t nwordsn, telr_seqset **out);

/* ---- FASTA / FASTQ text -> the arrays above (host code; plain files, not gzip).  Replaces handing the file names to
 *      ngmlr / minimap2 (src/telr/TELR_alignment.py:31-51, 69-82).  Names end at the first white space. */
typedef struct telr_fasta telr_fasta;
int  telr_fasta_load(const char *path, telr_fasta **out);
int32_t telr_fasta_count(const telr_fasta *f);
int64_t telr_fasta_bases(const telr_fasta *f);                  /* sum of the lengths */
int64_t telr_fasta_extent(const telr_fasta *f);                 /* bytes behind telr_fasta_seq that the offsets may point into */
const char *telr_fasta_seq(const telr_fasta *f);                 /* base buffer: sequence i = [off[i], off[i] + len[i]); the sequences are
                                                                    packed end to end, or -- when every sequence of the file sits on one line --
                                                                    the buffer is the mapped file itself and nothing was copied */
const int64_t *telr_fasta_off(const telr_fasta *f);
const int32_t *telr_fasta_len(const telr_fasta *f);
const char *const *telr_fasta_names(const telr_fasta *f);
void telr_fasta_free(telr_fasta *f);

/* ---- index (replaces "minimap2 ... REF" re-indexing REF on every call) ----- */
int  telr_index_build(telr_ctx *ctx, const telr_seqset *targets, const telr_idx_opt *io,
                      telr_index **out);
void telr_index_free(telr_index *idx);
/* number of minimizers / distinct minimizers, for reports */
int  telr_index_stats(const telr_index *idx, int64_t *n_minimizers, int64_t *n_distinct);

/* ---- mapping (S1,S2,S7: all queries vs all targets; S3,S4,S6: query i vs the
 *      single target qtarget[i]; S5: TELR_MF_PER_TARGET) ------------------------
 * qtarget may be NULL (every query sees every target) or hold one target id
 * per query (-1 = all).  Blocking; internally stream-asynchronous.
 * A query set of any size is accepted: up to 1.6 Gbp is one range; a larger one streams through in an even number of equal
 * ranges of at most 1.4 Gbp, two of them in flight (each bounded by an anchor budget at the density the index has shown), and the records
 * come back in query order whatever the cut.  The scratch is grow-only per context (about 75 B per query base of the
 * largest range at 0.25 anchors per base); TELR_E_NOMEM with two ranges in flight makes the call run again one range at a
 * time before it is reported. */
int  telr_map(telr_ctx *ctx, const telr_index *idx, const telr_seqset *queries,
              const int32_t *qtarget, const telr_map_opt *mo, telr_result **out);

int64_t         telr_result_count(const telr_result *r);
const telr_aln *telr_result_alns(const telr_result *r);     /* sorted by (qid, rank) */
/* CIGAR ops (BAM encoding, len<<4|op) of all records in one array; a record owns ops
 * [cigar_off, cigar_off + n_cigar).  The array is filled by one DMA that starts before the final
 * chain selection, so it may also hold the ops of chains that selection dropped (unreferenced).
 * telr_map returns as soon as the RECORDS are complete; the DMA of the CIGAR array may still be in flight
 * (so that a caller streaming batches overlaps it with the next telr_map call).  telr_result_cigars,
 * the writers, telr_depth_medians and telr_result_free wait for it; telr_result_wait does so explicitly. */
/* A result made of records and CIGAR words the caller holds (copied; cigar_off must index `cigars`): for the writers and
 * telr_depth_medians on records that were mapped elsewhere -- at N > 1 ranks the stage-1 hand-off is ONE sorted BAM, written by
 * rank 0 from the records every rank mapped (telr_amd/shard.py: gather_stage1). */
int             telr_result_from_arrays(telr_ctx *ctx, const telr_aln *alns, int64_t n, const uint32_t *cigars, int64_t n_cigar,
                                        telr_result **out);
/* the same with the CIGAR words in DEVICE memory (e.g. the receive buffer of an all-to-all): copied device-to-device into the
 * result's device array -- which the device BAM writer reads in place -- and mirrored to the host array once */
int             telr_result_from_device_cigars(telr_ctx *ctx, const telr_aln *alns, int64_t n, const void *d_cigars, int64_t n_cigar,
                                               telr_result **out);
int             telr_result_wait(const telr_result *r);
int64_t         telr_result_cigar_count(const telr_result *r);
const uint32_t *telr_result_cigars(const telr_result *r);
void            telr_result_free(telr_result *r);

/* ---- text emitters at the reference's own boundary (PAF for S4,S5,S7; SAM for S1,S2,S3,S6) --------
 * qnames / tnames: arrays of C strings indexed by query / target id.  path NULL = stdout. */
int  telr_write_paf(const telr_result *r, const char *const *qnames, const char *const *tnames, int with_cigar,
                    const char *path, int append);
#define TELR_SAM_MD          0x1   /* --MD */
#define TELR_SAM_CS          0x2   /* --cs */
#define TELR_SAM_SOFTCLIP    0x4   /* -Y  */
#define TELR_SAM_NO_UNMAPPED 0x8
#define TELR_SAM_PRIMARY_ONLY 0x10 /* drop secondary and supplementary records: `samtools view -F0x900`          */
#define TELR_SAM_SORTED      0x20  /* coordinate order (target, position; unmapped last): `samtools sort`         */
#define TELR_SAM_NO_HEADER   0x40  /* records only, as `samtools view` without -h.  PRIMARY_ONLY|SORTED|NO_HEADER =
                                      the text `samtools view -F0x900 sorted.bam` pipes into wtpoa-cns at the
                                      polishing site (hand-off H3, src/telr/TELR_assembly.py:208,228)                */
/* sequences are needed for SEQ, NM, MD and cs: concatenated ASCII + offsets + lengths as in telr_seqset_create.
 * rg_id NULL = no @RG line / RG tag (minimap2 sites); NGMLR site passes --rg-id/--rg-sm/--rg-lb. */
int  telr_write_sam(const telr_result *r, int32_t n_queries, const char *const *qnames, const char *q_ascii, const int64_t *q_off,
                    const int32_t *q_len, int32_t n_targets, const char *const *tnames, const char *t_ascii, const int64_t *t_off,
                    const int32_t *t_len, int32_t flags, const char *rg_id, const char *rg_sm, const char *rg_lb,
                    const char *pg_line, const char *path);

/* Coordinate-sorted BAM (+ .bai when write_index != 0): replaces `samtools sort -o BAM SAM; samtools index BAM`
 * (src/telr/TELR_alignment.py:103-114).  Same arguments as telr_write_sam; level = zlib level (0 -> 1). */
int  telr_write_bam(const telr_result *r, int32_t n_queries, const char *const *qnames, const char *q_ascii, const int64_t *q_off,
                    const int32_t *q_len, int32_t n_targets, const char *const *tnames, const char *t_ascii, const int64_t *t_off,
                    const int32_t *t_len, int32_t flags, const char *rg_id, const char *rg_sm, const char *rg_lb,
                    const char *pg_line, const char *bam_path, int32_t write_index, int32_t level);

/* The same file made on the DEVICE from what is already resident there (the reads, the reference, the CIGARs): NM / MD / cs /
 * SA, the 4-bit SEQ, the coordinate sort (refID, position, forward before reverse strand, then query order) and the BGZF
 * blocks (CRC-32 included) are computed by kernels, the host only moves the finished file image into `bam_path` and writes
 * the .bai.  `queries` = the set the result was mapped from, `idx` = the index it was mapped against (its targets supply the
 * reference bases).  No ASCII sequences are needed.  level 0 = stored BGZF blocks; level >= 1 = deflate blocks coded on
 * the device (Huffman tables per BAM field class, run-length matches).  Bases print as the engine sees them: A C G T, anything
 * else N (telr_write_bam / telr_write_sam print the same, so the uncompressed streams of the two writers are equal). */
/* Optional, before telr_map: start creating `bam_path` in the background -- the file is created and mapped at once, then
 * est_bytes of it (a BAM with --cs --MD takes about 0.85 bytes per read base at level 1, 2.8 at level 0) are allocated and
 * pre-faulted into the mapping block by block, front to back -- so that telr_write_bam_dev(... the same path ...) copies the
 * finished file image into pages that exist and are mapped (80+ GB/s instead of the 6-15 GB/s of a fresh page-cache page),
 * and cuts the file to size.  Without it, or past the estimate, the writer streams through a pinned ring and one pwrite
 * thread.  The mapping is taken apart by a background thread after the file is complete; telr_bam_release_wait() waits for
 * that (only a process that prepares another file right away has a reason to).  A prepared file that is never written is
 * removed from the context by the next telr_bam_prepare / telr_destroy / telr_bam_discard (the file itself stays: a caller
 * whose mapping failed unlinks it -- the reference tests for the file's existence, TELR_alignment.py:110-114).  A writer call
 * that fails removes `bam_path` and its .bai itself.  One telr_write_bam_dev runs at a time per process. */
/* ---- ONE coordinate-sorted BAM written by N ranks (SURVEY 8e x TELR_alignment.py:103-114; TELR_sv.py:35-47 reads one file) ----
 * The records of the job are range-partitioned by (refID, position); rank k holds the reads that have a record in slice k with
 * ALL their records (`emit[i]` = record i of the result lies in this slice; the others are only there so that the SA tag of an
 * emitted record can name them) and makes the BGZF blocks of its slice on its device: telr_write_bam_slice (with_header: rank 0;
 * TELR_SAM_NO_UNMAPPED in flags on every rank but the last, which also holds the reads without a record).  The slice's size is
 * then known (telr_bam_segment_info: out[0] bytes in the file, [1] mapped records, [2] unmapped reads, [3] uncompressed
 * bytes); after an exclusive scan of the sizes over the ranks every rank puts its image at its place of the one file
 * (telr_bam_segment_write: `path` exists, is_last appends the BGZF EOF block) and hands what the index needs of its records
 * (telr_bam_segment_entries: arrays of [1] entries in file order, virtual offsets for the slice at file_off) to rank 0, which
 * writes the one .bai from the concatenated arrays (telr_bai_write).  The inflated stream of the N-rank file equals the
 * one-rank file's byte for byte (same records, same order -- ties by the job-level read number: the reads of a rank must be
 * in that order); the BGZF block boundaries differ.  The image lives in the context until its next writer call. */
typedef struct telr_bam_segment telr_bam_segment;
int  telr_write_bam_slice(telr_ctx *ctx, const telr_result *r, const telr_seqset *queries, const telr_index *idx,
                          const char *const *qnames, const char *const *tnames, int32_t flags,
                          const char *rg_id, const char *rg_sm, const char *rg_lb, const char *pg_line,
                          const uint8_t *emit, int32_t with_header, int32_t level, telr_bam_segment **out);
int  telr_bam_segment_info(const telr_bam_segment *s, int64_t *out);
int  telr_bam_segment_entries(const telr_bam_segment *s, int64_t file_off, int32_t *tid, int32_t *ts, int32_t *te, uint64_t *vb, uint64_t *v_end);
int  telr_bam_segment_write(telr_ctx *ctx, const telr_bam_segment *s, const char *path, int64_t file_off, int32_t is_last);
void telr_bam_segment_free(telr_bam_segment *s);
int  telr_bai_write(const char *bai_path, int64_t n, const int32_t *tid, const int32_t *ts, const int32_t *te, const uint64_t *vb, uint64_t v_end,
                    int64_t n_unmapped, int32_t n_targets, const int32_t *t_len);
int  telr_bam_prepare(telr_ctx *ctx, const char *bam_path, int64_t est_bytes);
int  telr_bam_release_wait(void);
int  telr_bam_discard(telr_ctx *ctx);
int  telr_write_bam_dev(telr_ctx *ctx, const telr_result *r, const telr_seqset *queries, const telr_index *idx,
                        const char *const *qnames, const char *const *tnames, int32_t flags, const char *rg_id, const char *rg_sm,
                        const char *rg_lb, const char *pg_line, const char *bam_path, int32_t write_index, int32_t level);

/* ---- pile-up consensus of the index's targets from the PRIMARY records of a result (neither secondary nor supplementary:
 *      `samtools view -F0x900`).  Stands where the reference pipes that view into `wtpoa-cns -d CNS -i -` to polish a locus'
 *      draft contig (src/telr/TELR_assembly.py:226-247) -- with a different algorithm: a majority vote per contig position
 *      (base / deletion / up to 8 inserted bases after it; D runs longer than 30 do not vote; positions covered by fewer than
 *      min_depth records keep the draft base), not wtpoa-cns's partial-order alignment.  Offered behind telr_assembly's polish="pileup".  DESIGN.md 3.12. */
typedef struct telr_consensus telr_consensus;
int  telr_consensus_build(telr_ctx *ctx, const telr_result *r, const telr_seqset *queries, const telr_index *idx, int32_t min_depth,
                          telr_consensus **out);
/* The same hand-off with a WINDOW PARTIAL-ORDER consensus (opt-in like the pile-up: wtpoa-cns's own source is
 * absent, this is the published scheme of window POA polishing): the draft is cut into 200-base windows, every primary record
 * that covers a window whole gives the piece of its read its CIGAR aligns there, the pieces are re-aligned one by one to a graph
 * that starts as the draft's window (sequence-to-graph DP, one wave per window) and merged into it, the window's consensus is
 * the heaviest-bundle path.  Unlike the pile-up vote it re-phases columns: it does not trust the pairwise CIGARs inside a window. */
int  telr_poa_build(telr_ctx *ctx, const telr_result *r, const telr_seqset *queries, const telr_index *idx, int32_t min_depth,
                    telr_consensus **out);
int32_t telr_consensus_count(const telr_consensus *c);              /* = number of targets */
const char *telr_consensus_seq(const telr_consensus *c);             /* concatenated consensus sequences (A C G T N) */
const int64_t *telr_consensus_off(const telr_consensus *c);
const int32_t *telr_consensus_len(const telr_consensus *c);
void telr_consensus_free(telr_consensus *c);

/* ---- fused "samtools depth -aa -r | median" (D) --------------------------------
 * For n_iv intervals (target id, 0-based start, 0-based INCLUSIVE end — the
 * reference feeds 0-based numbers into samtools' 1-based inclusive region
 * string, TELR_te.py:870-884, so E-S+1 positions are read starting one base
 * to the left; callers pass the positions they want counted) compute the
 * per-base depth of the result's primary+supplementary records (secondary
 * skipped, deletions not counted) and return the median as samtools+python
 * `statistics.median` would (mean of the two middle values for even counts). */
int  telr_depth_medians(telr_ctx *ctx, const telr_result *r, int32_t n_targets,
                        const int32_t *target_len, int32_t n_iv, const int32_t *iv_tid,
                        const int32_t *iv_start, const int32_t *iv_end, double *median_out);

/* ---- window reads (a12) -------------------------------------------------------
 * Replaces the per-locus `pysam.AlignmentFile(bam).fetch(chr, bp-1000, bp+1000)` loop of prep_assembly_inputs
 * (src/telr/TELR_assembly.py:384-415, read_type="all"): for every window w = (win_tid, [win_lo, win_hi)) the ascending,
 * distinct query ids with ANY record (primary, secondary, supplementary) overlapping it: ts < win_hi and te > win_lo.
 * recs: any array of records (telr_result_alns of the stage-1 result, or records put together by the caller).  Host
 * code, no device needed.  out_off[n_win+1]; the ids of window w are out_qid[out_off[w] .. out_off[w+1]).  *needed =
 * total number of ids; when it exceeds cap nothing is copied and TELR_E_RANGE is returned: call again with cap >= *needed. */
int  telr_window_reads(const telr_aln *recs, int64_t n_rec, int32_t n_win, const int32_t *win_tid, const int32_t *win_lo,
                       const int32_t *win_hi, int64_t *out_off, int32_t *out_qid, int64_t cap, int64_t *needed);

/* ---- timing of the last telr_map / telr_index_build call (HIP events on the
 *      engine's own stream).  Stage names: telr_stage_name(i). ---------------- */
#define TELR_N_STAGES 16
int  telr_stage_ms(const telr_ctx *ctx, float *ms_out /* [TELR_N_STAGES] */);
const char *telr_stage_name(int i);
/* algorithmic work counters of the last telr_map call (SURVEY §8d terms) */
typedef struct telr_counters {
    int64_t query_bases, minimizers, probes, anchors, chains, dp_problems, dp_cells,
            window_bases, cigar_ops, records;
    /* engine-only (the oracle leaves them 0): queries with more anchors than one workgroup sorts in LDS (a read in a tandem
     * array / satellite), and the ranges that held at least one such query and therefore took the two-step seeding + the
     * library's segmented sort for it */
    int64_t over_queries, over_ranges;
} telr_counters;
int  telr_last_counters(const telr_ctx *ctx, telr_counters *out);
/* per DP class (TELR_N_DPCLS classes) of the last telr_map call:
 * out[c*4+0] problems, [c*4+1] DP cells, [c*4+2] anti-diagonal steps (sum of m+n),
 * [c*4+3] algorithmic bytes (2-bit bases read once + 4 B per CIGAR run + 32 B result) */
#define TELR_N_DPCLS 25
int  telr_last_dp_classes(const telr_ctx *ctx, int64_t *out /* [TELR_N_DPCLS*4] */);

#ifdef __cplusplus
}
#endif
#endif /* TELR_HIP_H */
