"""Read selection around candidate loci, mirroring the `read_type="all"` branch of the reference's
`prep_assembly_inputs` (src/telr/TELR_assembly.py:384-415): for every locus of the VCF table, every read
with ANY alignment record (primary, secondary or supplementary) overlapping
[breakpoint-1000, breakpoint+1000) on the locus chromosome, breakpoint = round((start+end)/2).

The reference goes through `pysam.AlignmentFile.fetch` on the sorted stage-1 BAM and then
seqtk / sort|uniq / csplit (:419-456); here the stage-1 records are still in memory, so the selection is a
masked lookup over the record arrays.  The reference iterates a Python `set` of read names, so its per-locus
read ORDER depends on PYTHONHASHSEED; here reads come out in input order (the count, column 14 of
`vcf_filtered.tsv.new`, is identical).
"""
import numpy as np


def breakpoint(start, end):
    return round((int(start) + int(end)) / 2)          # Python round: banker's rounding, as in the reference


def window_reads(alns, chrom_ids, loci, window=1000):
    """alns: record array of telr_map (fields qid, tid, ts, te); chrom_ids: {chromosome name: target id};
    loci: rows of the vcf table (chr, start, end, ...).  -> list of sorted unique read-index arrays."""
    tid = np.asarray(alns["tid"]); ts = np.asarray(alns["ts"]); te = np.asarray(alns["te"]); qid = np.asarray(alns["qid"])
    order = np.lexsort((ts, tid))
    tid_s, ts_s, te_s, qid_s = tid[order], ts[order], te[order], qid[order]
    out = []
    for row in loci:
        c = chrom_ids.get(row[0], -1)
        bp = breakpoint(row[1], row[2])
        s, e = max(0, bp - window), bp + window
        lo = np.searchsorted(tid_s, c, side="left"); hi = np.searchsorted(tid_s, c, side="right")
        if hi <= lo or c < 0:
            out.append(np.zeros(0, np.int64)); continue
        # records starting before e; among them those ending after s
        k = lo + np.searchsorted(ts_s[lo:hi], e, side="left")
        m = te_s[lo:k] > s
        out.append(np.unique(qid_s[lo:k][m]).astype(np.int64))
    return out


def annotate_vcf_with_counts(loci, reads_per_locus):
    """the `.new` copy of the table: every row + the number of window reads (column 14)"""
    return [list(r) + [str(len(x))] for r, x in zip(loci, reads_per_locus)]
