"""Stage-2 hand-offs around the local assembler (hand-off point H3), mirroring `src/telr/TELR_assembly.py`:

* `window_reads` / `annotate_vcf_with_counts`: read selection around candidate loci, mirroring the `read_type="all"` branch of the reference's
`prep_assembly_inputs` (src/telr/TELR_assembly.py:384-415): for every locus of the VCF table, every read
with ANY alignment record (primary, secondary or supplementary) overlapping
[breakpoint-1000, breakpoint+1000) on the locus chromosome, breakpoint = round((start+end)/2).

The reference goes through `pysam.AlignmentFile.fetch` on the sorted stage-1 BAM and then
seqtk / sort|uniq / csplit (:419-456); here the stage-1 records are still in memory, so the selection is a
masked lookup over the record arrays.  The reference iterates a Python `set` of read names, so its per-locus
read ORDER depends on PYTHONHASHSEED; here reads come out in input order (the count, column 14 of
`vcf_filtered.tsv.new`, is identical).

* `polish_alignments`: the aligner half of `run_wtdbg2_polishing` (:185-260): `minimap2 -t N -ax P -r2k CNS READS | samtools
sort` followed by `samtools view -F0x900 BAM | wtpoa-cns -d CNS -i -` (:199-236).  wtpoa-cns itself is a hand-off point and
is not built here; what it reads on stdin is.
"""
import os

import numpy as np


def breakpoint(start, end):
    return round((int(start) + int(end)) / 2)          # Python round: banker's rounding, as in the reference


def window_reads(alns, chrom_ids, loci, window=1000):
    """alns: record array of telr_map (fields qid, tid, ts, te); chrom_ids: {chromosome name: target id};
    loci: rows of the vcf table (chr, start, end, ...).  -> list of sorted unique read-index arrays.

    One call of the library's host routine `telr_window_reads` over the record array in place (the stage-1 records of a
    30x run are ~50 MB: no per-field copies, no sort of the records)."""
    import ctypes as C
    from . import _lib
    from ._abi import ALN_DTYPE
    n = len(loci)
    c_of = np.array([chrom_ids.get(row[0], -1) for row in loci], np.int32)
    bp = np.array([breakpoint(row[1], row[2]) for row in loci], np.int64)
    lo = np.maximum(0, bp - window).astype(np.int32); hi = (bp + window).astype(np.int32)
    if n == 0:
        return []
    if alns.dtype != ALN_DTYPE or not alns.flags["C_CONTIGUOUS"]:      # records from elsewhere (a BAM, a test): the four fields suffice
        full = np.zeros(len(alns), ALN_DTYPE)
        for f in ("qid", "tid", "ts", "te"):
            full[f] = alns[f]
        alns = full
    L = _lib.lib()
    off = np.zeros(n + 1, np.int64); need = C.c_int64(0)
    cap = max(1024, 96 * n)
    while True:
        qid = np.empty(cap, np.int32)
        rc = L.telr_window_reads(alns.ctypes.data, len(alns), n, c_of.ctypes.data, lo.ctypes.data, hi.ctypes.data, off.ctypes.data, qid.ctypes.data, cap, C.byref(need))
        if rc == 0:
            break
        if need.value <= cap:
            raise _lib.TelrError("telr_window_reads: " + L.telr_strerror(rc).decode())
        cap = int(need.value)
    qid = qid[:need.value].astype(np.int64)
    return [qid[off[i]:off[i + 1]] for i in range(n)]


def annotate_vcf_with_counts(loci, reads_per_locus):
    """the `.new` copy of the table: every row + the number of window reads (column 14)"""
    return [list(r) + [str(len(x))] for r, x in zip(loci, reads_per_locus)]


def polish_alignments(engine, contig_names, contig_seqs, reads_by_locus, read_names=None, presets="ont", tmp_path=None):
    """Site S3 for ALL loci in one engine call: the reads of locus k against its draft contig k (`-ax map-ont|map-pb -r2k`:
    band width 2000, SAM output), then per locus the text that `samtools sort | samtools view -F0x900` pipes into
    `wtpoa-cns -i -`: coordinate-sorted SAM records, primary alignments only (no 0x100 / 0x800), no header, unmapped
    reads last (`view -F0x900` keeps them; wtpoa-cns skips them).

    reads_by_locus[k]: read sequences (str) of locus k (SEQ is part of what wtpoa-cns reads, so the bases are needed on
    the host anyway).  read_names[k]: their names (default "<contig>_r<i>").  -> (list of SAM texts per locus, record array, cigar array)"""
    import os
    import tempfile
    from .presets import preset
    io, mo = preset("map-pb" if presets == "pacbio" else "map-ont")
    mo.bw = 2000                                            # -r2k (TELR_assembly.py:205)
    ix = engine.index(list(contig_seqs), io)
    qt, qnames, flat = [], [], []
    for k, rs in enumerate(reads_by_locus):
        for i, r in enumerate(rs):
            qt.append(k); flat.append(r)
            qnames.append(read_names[k][i] if read_names is not None else "%s_r%d" % (contig_names[k], i))
    if not flat:
        return ["" for _ in contig_names], np.zeros(0), np.zeros(0, np.uint32)
    q_host = list(flat)
    r = ix.map_raw(engine.seqset(q_host), mo, qtarget=np.asarray(qt, np.int32))
    try:
        res = ix.result_arrays(r)
        fd, path = tempfile.mkstemp(suffix=".sam", dir=tmp_path)
        os.close(fd)
        try:
            # The text is written with the query / locus INDEX as QNAME / RNAME and the real names are put back here: the
            # reference commonly puts one read into the windows of two neighbouring loci, and two loci may carry the same
            # contig name -- a line belongs to the locus its query was confined to, whatever it is called.
            ix.write_sam(r, [str(i) for i in range(len(qnames))], q_host, [str(k) for k in range(len(contig_names))], list(contig_seqs), path,
                         md=False, cs=False, softclip=False, primary_only=True, coordinate_sorted=True, header=False)
            mapped = [[] for _ in contig_names]
            unmapped = [[] for _ in contig_names]
            with open(path) as fh:
                for line in fh:
                    f = line.split("\t", 3)
                    qi = int(f[0]); k = qt[qi]
                    if f[2] == "*":
                        unmapped[k].append("%s\t%s\t*\t%s" % (qnames[qi], f[1], f[3]))
                    else:
                        rest = f[3]
                        if "\tSA:Z:" in rest:        # the other records of the read lie on the same contig (the query is confined to it)
                            head, sa = rest.split("\tSA:Z:", 1)
                            sa, tail = (sa.split("\t", 1) + [""])[:2] if "\t" in sa else (sa.rstrip("\n"), "\n")
                            sa = ";".join((contig_names[k] + x[x.index(","):]) if x else x for x in sa.split(";"))
                            rest = head + "\tSA:Z:" + sa + ("\t" + tail if tail != "\n" else "\n")
                        mapped[k].append("%s\t%s\t%s\t%s" % (qnames[qi], f[1], contig_names[k], rest))
        finally:
            os.remove(path)
    finally:
        ix.free_raw(r)
    return ["".join(mapped[k]) + "".join(unmapped[k]) for k in range(len(contig_names))], res.alns, res.cigars


def polish_consensus(engine, contig_names, contig_seqs, reads_by_locus, presets="ont", iterations=1, min_depth=3, read_set=None, method="pileup", timings=None):
    """The polishing loop of `run_wtdbg2_polishing` (TELR_assembly.py:185-262) with the consensus made on the device: per
    iteration ONE engine call maps the reads of every locus to its draft contig (`-ax P -r2k`, as S3) and ONE pile-up pass over
    the primary records (`-F0x900`) rewrites all contigs (`telr_consensus_build`: majority vote per position, spec 3.12).
    This is NOT wtpoa-cns's partial-order alignment -- a different consensus algorithm can change call sets, which is why the
    locus pipeline only uses it on request (`polish="pileup"`).  reads_by_locus[k]: read sequences, or -- with `read_set`, the
    stage-1 SeqSet resident on the device -- read indices (gathered on the device, as in telr_af.get_af).
    method="poa": the window partial-order consensus instead (`telr_poa_build`, spec 3.13: the reads are re-aligned to a graph of
    each 200-base window; closer to what wtpoa-cns does, still not its code).
    timings: a dict that receives the wall-clock seconds of the pass's phases (read set, index, map, consensus).
    -> list of polished contig sequences (a contig no read maps to stays as it is)."""
    if method not in ("pileup", "poa"):
        raise ValueError("method must be 'pileup' or 'poa'")
    counts_all = np.fromiter((len(rs) for rs in reads_by_locus), np.int64, len(reads_by_locus))
    # Round 6: on the engine the loci are polished as TWO halves at a time, the second one on the engine's second context in a host thread of its
    # own (per-locus results do not depend on what else is in the call: per-query targets, windows per contig): the host work of a half -- packing,
    # the range plan, the pieces of the window consensus from the CIGARs, grouping, uploads, Python -- runs under the other half's kernels
    # (configs[2], window consensus: 164-184 -> see DESIGN section 6).  TELR_POLISH_HALVES=1 keeps one call (A/B; tests hold both to the same strings).
    two = hasattr(engine, "worker") and len(reads_by_locus) >= 64 and int(counts_all.sum()) > 0 and os.environ.get("TELR_POLISH_HALVES", "2") != "1"
    if two:
        fr, tot = engine.mem_info()
        two = fr >= 0.3 * tot and not getattr(engine.worker(), "crowded", False)
    if two:
        import threading
        cum = np.cumsum(counts_all)
        cut = int(np.searchsorted(cum, cum[-1] // 2)) + 1
        cut = min(max(cut, 1), len(reads_by_locus) - 1)
        out = [None, None]; err = [None]
        t_b = {} if timings is not None else None
        def second():
            try:
                out[1] = _polish_part(engine.worker(), contig_seqs[cut:], reads_by_locus[cut:], presets, iterations, min_depth, read_set, method, t_b)
            except Exception as e:            # no room beside the first half (or any other failure of the second context): that half in turn, below
                err[0] = e
        th = threading.Thread(target=second); th.start()
        try:
            out[0] = _polish_part(engine, contig_seqs[:cut], reads_by_locus[:cut], presets, iterations, min_depth, read_set, method, timings)
        finally:
            th.join()
        if err[0] is not None:
            if "out of device memory" not in str(err[0]) and "out of memory" not in str(err[0]):
                raise err[0]
            engine.worker().release_scratch()
            out[1] = _polish_part(engine, contig_seqs[cut:], reads_by_locus[cut:], presets, iterations, min_depth, read_set, method, timings)
        elif timings is not None:
            timings["second_half_s"] = sum(t_b.values())
        return out[0] + out[1]
    return _polish_part(engine, contig_seqs, reads_by_locus, presets, iterations, min_depth, read_set, method, timings)


def _polish_part(engine, contig_seqs, reads_by_locus, presets, iterations, min_depth, read_set, method, timings):
    """polish_consensus for the loci given, one engine call per iteration on `engine`"""
    import time
    def _t(key, t0):
        if timings is not None:
            timings[key] = timings.get(key, 0.0) + time.time() - t0
        return time.time()
    if method not in ("pileup", "poa"):
        raise ValueError("method must be 'pileup' or 'poa'")
    from .presets import preset
    io, mo = preset("map-pb" if presets == "pacbio" else "map-ont")
    mo.bw = 2000
    if hasattr(engine, "worker"):                    # the engine: the pile-up reads the CIGARs where they are made (TELR_MF_KEEP_CIGARS)
        from ._abi import MF_KEEP_CIGARS
        mo = mo.copy(); mo.flags |= MF_KEEP_CIGARS
    t0 = time.time()
    contigs = [c if isinstance(c, str) else bytes(c).decode() for c in contig_seqs]
    counts = np.fromiter((len(rs) for rs in reads_by_locus), np.int64, len(reads_by_locus))
    if int(counts.sum()) == 0:
        return contigs
    qt = np.repeat(np.arange(len(reads_by_locus), dtype=np.int32), counts)           # the locus of every read, in locus order
    if read_set is not None:                                                         # read indices: no Python loop over 40 k reads
        flat = np.concatenate([np.asarray(rs, np.int32).reshape(-1) for rs in reads_by_locus])
        qset = read_set.subset(flat, eng=engine) if hasattr(read_set, "eng") else read_set.subset(flat)      # (the gather on THIS context's stream: contexts are not re-entrant)
    else:
        qset = engine.seqset([r for rs in reads_by_locus for r in rs])
    t0 = _t("read_set_s", t0)
    for _ in range(max(1, int(iterations))):
        ix = engine.index(contigs, io)
        t0 = _t("index_s", t0)
        r = ix.map_raw(qset, mo, qtarget=qt)
        t0 = _t("map_s", t0)
        if os.environ.get("TELR_AF_TRACE") and hasattr(engine, "dp_classes"):
            import sys
            sys.stderr.write("polish map: stages " + ", ".join("%s %.1f" % kv for kv in engine.stage_ms().items() if kv[1] > 0.5) + " | dp classes (problems, Mcells): " +
                             ", ".join("%d: %d %.0f" % (c, v[0], v[1] / 1e6) for c, v in enumerate(np.asarray(engine.dp_classes()).reshape(-1, 4).tolist()) if v[0]) + "\n")
        try:
            contigs = ix.consensus(r, qset, min_depth=min_depth, poa=method == "poa")
            t0 = _t("consensus_s", t0)
        finally:
            ix.free_raw(r)
            ix.free()
    qset.free()
    _t("free_s", t0)
    return contigs
