"""Allele-frequency estimation from read->contig realignments (stage 4 of the path).

Restates `get_af` and its helpers of the reference (`src/telr/TELR_te.py`: `realignment` :495-515,
`get_flank_cov` :518-550, `get_te_flank_ratio` :564-575, `get_af` :578-838, `get_te_cov` :841-867,
`get_median_cov` :870-884).  The reference spawns, per locus and per contig orientation, one
`minimap2 -a` + three samtools processes and eight `samtools depth` processes; here all loci are
mapped in two batched engine calls (forward and reverse-complement contigs are just two targets
of one index) and the per-base depth + medians run on the device (`telr_depth_medians`).

Quirks kept (pinned by tests/golden/af_*.json, captured from the reference's own code):
  * only the 5' TE/flank medians of the forward and of the reverse-complement run enter the
    frequency (:810-817); the 3' medians are reported but unused;
  * a median of 0 counts as missing (`if te_cov and flank_cov`, :565), ratios > 1.5 are dropped,
    the two sides must agree within 0.3, the result is capped at 1 and rounded to 3 digits;
  * the whole TE is used when start+offset+interval >= end (:849-866);
  * `samtools depth -r chr:S-E` is fed 0-based numbers although the region syntax is 1-based
    inclusive (:870-884): E-S+1 positions are read, shifted one base to the left.
"""
import numpy as np


def depth_region(start, end):
    """`samtools depth -aa -r chr:start-end` -> 0-based inclusive (first, last) positions it prints."""
    first = start - 1 if start > 0 else 0
    return first, end - 1


def te_cov_intervals(start, end, te_interval_size, te_offset):
    """-> [(s,e) for te_5p, (s,e) for te_3p] as passed to get_median_cov (:841-867)"""
    if te_interval_size and start + te_offset + te_interval_size < end:
        return [(start + te_offset, start + te_offset + te_interval_size),
                (end - te_interval_size - te_offset, end - te_offset)]
    return [(start, end), (start, end)]


def flank_cov_intervals(contig_length, start, end, flank_len, offset):
    """-> [left or None, right or None] (:518-550)"""
    left = right = None
    if start - flank_len - offset >= 0:
        left = (start - flank_len - offset, start - offset)
    if end + flank_len + offset <= contig_length:
        right = (end + offset, end + flank_len + offset)
    return [left, right]


def get_te_flank_ratio(te_cov, flank_cov):
    if te_cov and flank_cov:
        ratio = te_cov / flank_cov
        return None if ratio > 1.5 else ratio
    return None


def combine_af(te_5p_cov, flank_5p_cov, te_5p_cov_rc, flank_5p_cov_rc):
    """the inline frequency arithmetic of get_af (:810-835)"""
    taf_5p = get_te_flank_ratio(te_5p_cov, flank_5p_cov)
    taf_3p = get_te_flank_ratio(te_5p_cov_rc, flank_5p_cov_rc)
    if taf_5p and taf_3p:
        freq = (taf_5p + taf_3p) / 2 if abs(taf_5p - taf_3p) <= 0.3 else None
    elif taf_5p:
        freq = taf_5p
    elif taf_3p:
        freq = taf_3p
    else:
        freq = None
    if freq and freq > 1:
        freq = 1
    return round(freq, 3) if freq else None


_COMP_NP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTUNacgtun", b"TGCAANtgcaan"):
    _COMP_NP[_a] = _b


_COMP_BYTES = bytes(_COMP_NP)


def locus_intervals(start, end, contig_length, flank_interval, flank_offset, te_interval, te_offset):
    """The 8 depth queries of one locus: forward then reverse-complement, each te_5p, te_3p, flank_5p, flank_3p."""
    out = {}
    for tag, (s, e) in (("fw", (start, end)), ("rc", (contig_length - end, contig_length - start))):
        out[tag] = te_cov_intervals(s, e, te_interval, te_offset) + flank_cov_intervals(contig_length, s, e, flank_interval, flank_offset)
    return out


def locus_intervals_batch(start, end, length, flank_interval, flank_offset, te_interval, te_offset):
    """locus_intervals for arrays of loci at once -> (lo, hi, valid), each [n, 2, 4]: orientation (fw, rc) x (te_5p, te_3p,
    flank_5p, flank_3p); the same numbers as the scalar functions above give, locus by locus (tests/test_golden_glue.py)"""
    start = np.asarray(start, np.int64); end = np.asarray(end, np.int64); length = np.asarray(length, np.int64)
    n = len(start)
    lo = np.zeros((n, 2, 4), np.int64); hi = np.zeros((n, 2, 4), np.int64); valid = np.ones((n, 2, 4), bool)
    for o, (s, e) in enumerate(((start, end), (length - end, length - start))):
        inner = (s + te_offset + te_interval < e) if te_interval else np.zeros(n, bool)
        lo[:, o, 0] = np.where(inner, s + te_offset, s); hi[:, o, 0] = np.where(inner, s + te_offset + te_interval, e)
        lo[:, o, 1] = np.where(inner, e - te_interval - te_offset, s); hi[:, o, 1] = np.where(inner, e - te_offset, e)
        lo[:, o, 2] = s - flank_interval - flank_offset; hi[:, o, 2] = s - flank_offset; valid[:, o, 2] = lo[:, o, 2] >= 0
        lo[:, o, 3] = e + flank_offset; hi[:, o, 3] = e + flank_interval + flank_offset; valid[:, o, 3] = hi[:, o, 3] <= length
    return lo, hi, valid


def _median_or_none(x):
    """statistics.median returns an int for odd counts and a float for even counts; str() of it is
    re-parsed with float() in the reference, so only the value matters."""
    return None if x is None or (isinstance(x, float) and np.isnan(x)) else float(x)


def freq_table(medians):
    """medians: dict with the 8 values (None allowed) -> the reference's te_freq[contig] dict (9 keys)"""
    d = {
        "te_5p_cov": _median_or_none(medians["fw"][0]), "te_3p_cov": _median_or_none(medians["fw"][1]),
        "flank_5p_cov": _median_or_none(medians["fw"][2]), "flank_3p_cov": _median_or_none(medians["fw"][3]),
        "te_5p_cov_rc": _median_or_none(medians["rc"][0]), "te_3p_cov_rc": _median_or_none(medians["rc"][1]),
        "flank_5p_cov_rc": _median_or_none(medians["rc"][2]), "flank_3p_cov_rc": _median_or_none(medians["rc"][3]),
    }
    d["freq"] = combine_af(d["te_5p_cov"], d["flank_5p_cov"], d["te_5p_cov_rc"], d["flank_5p_cov_rc"])
    return d


class AfJob:
    """the S6 mapping of a set of loci, possibly still running in a host thread (af_start)"""
    def __init__(self):
        self.thread = None; self.exc = None; self.names = []; self.tindex = {}; self.ix = None; self.r = None; self.qs = None; self.lens = {}
        self.engine = None; self.threaded = False

    def wait(self):
        if self.thread is not None:
            self.thread.join(); self.thread = None
        if self.exc is not None:
            e, self.exc = self.exc, None
            self.release()
            raise e

    def release(self):
        if self.thread is not None:
            self.thread.join(); self.thread = None
        if self.r is not None and self.ix is not None:
            self.ix.free_raw(self.r)
        self.r = None
        if self.qs is not None and hasattr(self.qs, "free"):
            self.qs.free()
        if self.ix is not None and hasattr(self.ix, "free"):
            self.ix.free()
        self.qs = None; self.ix = None
        # a second context's scratch is grow-only like the first one's: on a device that is nearly full (configs[3] / [4]: the
        # stage-1 context alone holds 150-250 GB) it goes back at once, otherwise it stays for the next bundle
        if self.threaded and self.engine is not None and hasattr(self.engine, "mem_info"):
            eng, self.engine = self.engine, None
            try:
                fr, tot = eng.mem_info()
                if fr < 0.25 * tot:
                    eng.release_scratch()
                    eng.crowded = True             # run_loci: the next bundle runs S6 in turn (sizing this scratch again costs more than the overlap gains)
            except Exception:
                pass


def _af_pack(job, contigs):
    """-> (buffer, offsets, lengths) of the targets: every contig forward and reverse-complemented"""
    names = job.names
    # targets: contig k forward at 2k, reverse-complemented at 2k+1, as ONE byte buffer.  The reverse complement of the
    # concatenation of all contigs is the concatenation of their reverse complements in reverse order: one table lookup and
    # one reversal for the whole set instead of a translate + slice + decode per contig
    fw = [contigs[n].encode() if isinstance(contigs[n], str) else bytes(contigs[n]) for n in names]
    lens = np.array([len(b) for b in fw], np.int64)
    joined = b"".join(fw)
    cat = np.frombuffer(joined, np.uint8)
    rc_cat = np.frombuffer(joined.translate(_COMP_BYTES)[::-1], np.uint8)
    ends = np.cumsum(lens); starts = ends - lens; total = int(ends[-1])
    tlen = np.repeat(lens, 2)
    toff = np.zeros(2 * len(names), np.int64); toff[1:] = np.cumsum(tlen)[:-1]
    tbuf = np.empty(2 * total, np.uint8)
    for k in range(len(names)):               # two slice copies per contig (a gather over all bases at once is 50x slower)
        L = int(lens[k]); o = int(toff[2 * k])
        tbuf[o:o + L] = cat[starts[k]:ends[k]]
        tbuf[o + L:o + 2 * L] = rc_cat[total - ends[k]:total - starts[k]]
    return tbuf, toff, tlen.astype(np.int32)


def _af_map(job, engine, targets, reads_by_locus, presets, read_set):
    """index of the targets, window reads twice, ONE engine call"""
    from .presets import preset
    import os, time
    trace = os.environ.get("TELR_AF_TRACE"); t0 = time.time(); marks = []
    io, mo = preset("map-ont" if presets == "ont" else "map-pb")
    names = job.names
    job.ix = engine.index(targets, io)
    marks.append(("index", time.time() - t0))
    counts = np.array([len(reads_by_locus[n]) for n in names], np.int64)
    qtarget_fw = np.repeat(np.arange(len(names), dtype=np.int32) * 2, counts)
    # ONE engine call for both orientations: every read appears twice, once confined to the forward contig of its locus
    # and once to the reverse-complement contig (the reference runs two minimap2 jobs per locus, TELR_te.py:644-646)
    if read_set is not None:
        idx = np.concatenate([np.asarray(reads_by_locus[n], np.int32) for n in names]) if len(names) else np.zeros(0, np.int32)
        job.qs = read_set.subset(np.concatenate([idx, idx]), eng=engine) if hasattr(read_set, "eng") else read_set.subset(np.concatenate([idx, idx]))
    else:
        queries = [r for n in names for r in reads_by_locus[n]]
        job.qs = engine.seqset(queries + queries)
    qtarget_all = np.concatenate([qtarget_fw, qtarget_fw + 1])
    marks.append(("subset", time.time() - t0))
    if hasattr(engine, "worker"):                    # the engine: the depth step reads the CIGARs where they are made (TELR_MF_KEEP_CIGARS)
        from ._abi import MF_KEEP_CIGARS
        mo = mo.copy(); mo.flags |= MF_KEEP_CIGARS
    job.r = job.ix.map_raw(job.qs, mo, qtarget=qtarget_all)
    marks.append(("map", time.time() - t0))
    if trace:
        import sys
        sys.stderr.write("af_map: " + ", ".join("%s %.1f ms" % (k, v * 1e3) for k, v in marks) + " | engine stages " +
                         ", ".join("%s %.1f" % kv for kv in engine.stage_ms().items() if kv[1] > 0.5) + "\n")
        try:
            dc = engine.dp_classes()
            sys.stderr.write("af_map dp classes (problems, Mcells): " + ", ".join("%s: %d %.0f" % (c, v[0], v[1] / 1e6) for c, v in enumerate(np.asarray(dc).reshape(-1, 4).tolist()) if v[0]) + " | counters " + str({k: v for k, v in engine.counters().items() if v}) + "\n")
        except Exception as e:
            sys.stderr.write("af_map: no class counters (%s)\n" % e)


def af_start(engine, contigs, reads_by_locus, presets="ont", read_set=None, names=None, threaded=False, contig_set=None):
    """Start the S6 realignment (window reads -> forward and reverse-complement contig of their locus) of the loci `names`
    (default: every locus that has a contig and reads).  It needs no annotation: with threaded=True it runs in a host thread
    on `engine` (which must then be a context nothing else uses meanwhile: Engine.worker()) while the caller annotates and
    lifts the same loci.  contig_set = (SeqSet of contigs resident on the device, {locus name: its index there}): the forward /
    reverse-complement target set is then made on the device (SeqSet.subset(rc=...)) instead of being packed here and uploaded.
    -> AfJob for af_finish."""
    job = AfJob()
    job.engine = engine; job.threaded = threaded
    job.names = [n for n in (names if names is not None else contigs) if n in contigs and n in reads_by_locus]
    job.tindex = {n: 2 * k for k, n in enumerate(job.names)}
    job.lens = {n: len(contigs[n]) for n in job.names}
    if not job.names:
        return job
    if contig_set is not None:
        cset, where = contig_set
        order = np.array([where[n] for n in job.names], np.int32)
        targets = cset.subset(np.repeat(order, 2), eng=engine, rc=np.tile(np.array([0, 1], np.uint8), len(order)))
    else:
        targets = _af_pack(job, contigs)          # in the caller's thread: Python-bound work gains nothing from a second thread
    if not threaded:
        _af_map(job, engine, targets, reads_by_locus, presets, read_set)
        return job
    import threading

    def run():
        try:
            _af_map(job, engine, targets, reads_by_locus, presets, read_set)
        except BaseException as e:          # re-raised by wait()
            job.exc = e
    job.thread = threading.Thread(target=run, name="telr-af-map"); job.thread.start()
    return job


def af_finish(job, contig_te, flank_interval=100, flank_offset=200, te_interval=50, te_offset=50):
    """depth medians of the 8 intervals per annotated locus (contig_te: {locus: (start, end)}) on the job's records
    + the frequency arithmetic -> {locus name: te_freq dict}"""
    try:
        names = [n for n in job.names if n in contig_te]
        if not names:
            job.wait()
            return {}
        # the 8 depth queries of every locus, laid out while the mapping may still be running
        te = np.array([contig_te[n] for n in names], np.int64).reshape(-1, 2)
        lens = np.array([job.lens[n] for n in names], np.int64)
        lo, hi, valid = locus_intervals_batch(te[:, 0], te[:, 1], lens, flank_interval, flank_offset, te_interval, te_offset)
        first = np.where(lo > 0, lo - 1, 0); last = hi - 1                     # depth_region: samtools' 1-based region fed 0-based numbers
        tid = np.array([job.tindex[n] for n in names], np.int64)[:, None, None] + np.arange(2)[None, :, None] + np.zeros((1, 1, 4), np.int64)
        sel = np.nonzero(valid.reshape(-1))[0]
        job.wait()
        med_all = np.full(valid.size, np.nan)
        if len(sel):
            med_all[sel] = job.ix.depth_medians(job.r, tid.reshape(-1)[sel], first.reshape(-1)[sel], last.reshape(-1)[sel])
        # the tables of all loci: the values leave numpy ONCE (tolist gives Python floats, the same numbers float() would) instead of
        # eight isnan / float() calls per locus -- this loop runs after S6 has finished, i.e. on the bundle's critical path
        flat = med_all.reshape(len(names), 8)
        vals = flat.tolist(); nans = np.isnan(flat).tolist()
        out = {}
        for k, n in enumerate(names):
            v, z = vals[k], nans[k]
            m = [None if z[i] else v[i] for i in range(8)]
            d = {"te_5p_cov": m[0], "te_3p_cov": m[1], "flank_5p_cov": m[2], "flank_3p_cov": m[3],
                 "te_5p_cov_rc": m[4], "te_3p_cov_rc": m[5], "flank_5p_cov_rc": m[6], "flank_3p_cov_rc": m[7]}
            d["freq"] = combine_af(m[0], m[2], m[4], m[6])          # (= freq_table, which the golden tests pin)
            out[n] = d
        return out
    finally:
        job.release()


def get_af(engine, contigs, contig_te, reads_by_locus, presets="ont", flank_interval=100, flank_offset=200,
           te_interval=50, te_offset=50, read_set=None):
    """Batched replacement of get_af (:578-838).

    contigs: {locus name: contig sequence}; contig_te: {locus name: (start, end)} TE coordinates on the
    forward contig; reads_by_locus: {locus name: [read sequences]} (the +-1 kb window reads selected
    by prep_assembly_inputs(read_type="all"), TELR_assembly.py:384-462) or, with `read_set` (the stage-1
    SeqSet already resident on the device), {locus name: [read indices]}: the window reads are then gathered on
    the device instead of being packed and uploaded again.
    Returns {locus name: te_freq dict}.  (= af_start on the annotated loci + af_finish.)
    """
    job = af_start(engine, contigs, reads_by_locus, presets, read_set, names=[n for n in contig_te])
    return af_finish(job, contig_te, flank_interval, flank_offset, te_interval, te_offset)
