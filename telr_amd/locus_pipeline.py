"""The per-locus bundle of the path (stages 3-4) as batched engine calls:
   S4 ALT->contig, S5 library->contig, S7 flanks->reference + liftover decision tree, S6 reads->contig
   forward / reverse-complement + depth medians + allele frequency.
A "locus" = (name "<chr>_<start>_<end>", contig sequence, Sniffles ALT sequence, window reads)."""
from . import telr_te, telr_af, telr_liftover


def run_loci(backend, *args, **kw):
    """run_loci_impl with the device frees of the bundle's temporary sets and indexes held back to its end (aligner.deferred_frees)"""
    if hasattr(backend, "worker"):
        from .aligner import deferred_frees
        with deferred_frees():
            return run_loci_impl(backend, *args, **kw)
    return run_loci_impl(backend, *args, **kw)


def run_loci_impl(backend, ref_index, ref_names, ref_seq, loci, lib_names, lib_seqs, presets="ont", ref_te_rows=None,
             flank_len=500, gap=20, overlap=20, af_params=(100, 200, 50, 50), read_set=None, polish=None, polish_iterations=1, overlap_af=True):
    """loci: list of dicts(name, contig, alt, reads).  With `read_set` (the stage-1 read SeqSet resident on the
    device) a locus gives `read_idx` (indices into it) instead of `reads`.  polish="pileup": the draft contigs are first
    polished on the device with the locus' reads (telr_assembly.polish_consensus: the polishing loop of
    TELR_assembly.py:185-262 with a pile-up consensus in the place of wtpoa-cns -- a different algorithm, hence opt-in);
    polish="poa": the same with the window partial-order consensus (spec 3.13).
    -> dict(annotation, liftover, summary, af[, contigs])"""
    if polish in ("pileup", "poa") and loci:
        from . import telr_assembly
        rs = [l["read_idx"] if read_set is not None else l["reads"] for l in loci]
        pol = telr_assembly.polish_consensus(backend, [l["name"] for l in loci], [l["contig"] for l in loci], rs, presets=presets,
                                             iterations=polish_iterations, read_set=read_set, method=polish)
        loci = [dict(l, contig=c) for l, c in zip(loci, pol)]
    elif polish not in (None, "", "none"):
        raise ValueError("polish must be None, 'pileup' or 'poa'")
    names = [l["name"] for l in loci]
    contigs = {l["name"]: l["contig"] for l in loci}
    reads_by_locus = {l["name"]: (l["read_idx"] if read_set is not None else l["reads"]) for l in loci}
    # S6 (window reads -> forward / reverse-complement contig) needs the contigs only: on an engine with a second context it
    # runs in a host thread of its own while S4, S5, S7 and the liftover tree run here; the annotation meets it at the depth step
    job = None
    # on the engine the contigs are packed and uploaded ONCE: S4 / S5 index that set as it is, S6 takes its forward + reverse-complement
    # target set from it on the device
    cset = backend.seqset([l["contig"] for l in loci]) if loci and hasattr(backend, "worker") else None
    cwhere = {l["name"]: k for k, l in enumerate(loci)} if cset is not None else None
    if cwhere is not None and len(cwhere) != len(loci):
        cset = cwhere = None                    # (duplicate locus names: contigs[name] is the last one, the set's order would not say so)
    import os
    if overlap_af and loci and hasattr(backend, "worker") and not os.environ.get("TELR_LOCI_NO_OVERLAP"):          # (the switch: A/B runs)
        # a second context brings its own scratch: only where the device has room for it next to what stage 1 left behind
        fr, tot = backend.mem_info()
        if fr >= 0.3 * tot and not getattr(backend.worker(), "crowded", False):
            job = telr_af.af_start(backend.worker(), contigs, reads_by_locus, presets, read_set, threaded=True,
                                   contig_set=(cset, cwhere) if cset is not None else None)
    try:
        ann, s2c, t2c = telr_te.annotate_contig(backend, names, [l["contig"] for l in loci], [l["alt"] for l in loci],
                                                lib_names, lib_seqs, presets, contig_set=cset)
        mapper = telr_liftover.engine_flank_mapper(ref_index, ref_names)
        reports, summary = telr_liftover.liftover(mapper, contigs, ann, ref_seq, ref_te_rows, flank_len, gap, overlap)
        contig_te = {}
        for r in ann:                       # one annotation per contig feeds the AF step (first one wins, as a dict would)
            contig_te.setdefault(r[0], (int(r[1]), int(r[2])))
        freqs = None
        if job is not None:
            j, job = job, None
            try:
                freqs = telr_af.af_finish(j, contig_te, *af_params)
                freqs = {n: freqs[n] for n in contig_te if n in freqs}          # in annotation order, as get_af returns them
            except Exception as e:
                # the second context found no room next to the first one's scratch (a device filled by stage 1): give back what it
                # holds and run S6 in turn on this context, which takes back its own scratch when it has to
                if "out of device memory" not in str(e) and "out of memory" not in str(e):
                    raise
                backend.worker().release_scratch()
        if freqs is None:
            freqs = telr_af.get_af(backend, contigs, contig_te, reads_by_locus, presets, *af_params, read_set=read_set)
    finally:
        if job is not None:
            job.release()
    out = {"annotation": ann, "liftover": reports, "summary": summary, "af": freqs}
    if polish in ("pileup", "poa"):
        out["contigs"] = contigs
    return out


def locus_of_report(r):
    """locus (= contig) name of a liftover report: its ID is <contig>_<te_start>_<te_end> and contig names are
    <chromosome>_<start>_<end>, where the chromosome itself may contain underscores (chrUn_CP007071v1)"""
    return r["ID"].rsplit("_", 2)[0]


def locus_cost(l):
    """LPT cost of a locus: contig + ALT + window-read bases (whatever of them is known when loci are dealt)"""
    c = len(l["contig"]) + len(l.get("alt") or "")
    if "reads" in l:
        c += sum(len(r) for r in l["reads"])
    elif "read_bases" in l:
        c += int(l["read_bases"])
    return c


def run_loci_distributed(backend, ref_index, ref_names, ref_seq, loci, lib_names, lib_seqs, dist=None, device=None, shards=None, timings=None, **kw):
    """Stages 3-4 sharded over the ranks of one node: loci are assigned by LPT on their bases
    (`shards` = the per-rank lists of locus indices when the caller dealt them already, identical on every rank), every rank runs the bundle on its shard,
    and ONE all-gather of fixed-capacity blocks of fixed-width rows merges the coordinate / allele-frequency table
    (RCCL over xGMI on GPUs; gloo in the CPU tests).  Variable-length payloads (reports, sequences) stay on the
    owning rank.  -> (merged LOCUS_ROW array sorted by locus id, this rank's full results)"""
    from . import shard
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if shards is None:
        shards = shard.shard_loci([locus_cost(l) for l in loci], world)
    mine = shards[rank]
    capacity = max(len(x) for x in shards)          # every rank computes the same number: no size exchange
    sub = [loci[i] for i in mine]
    import time
    t0 = time.time()
    res = run_loci(backend, ref_index, ref_names, ref_seq, sub, lib_names, lib_seqs, **kw) if sub else \
        {"annotation": [], "liftover": [], "summary": {}, "af": {}}
    by_name = {}
    for r in res["liftover"]:
        by_name.setdefault(locus_of_report(r), r)
    ids, reps, freqs = [], [], []
    for gi, l in zip(mine, sub):
        r = by_name.get(l["name"])
        if r is None:                       # the locus did not pass annotation (no TE found on its contig): no row
            continue
        ids.append(gi); reps.append(r); freqs.append(res["af"].get(l["name"]))
    chrom_ids = {n: i for i, n in enumerate(ref_names)}
    fam_ids = {n: i for i, n in enumerate(lib_names)}
    rows = shard.rows_from_reports(ids, reps, freqs, chrom_ids, fam_ids)
    t1 = time.time()
    if timings is not None and world > 1:   # measured runs: how long this rank waits for the slowest one is not the collective's cost
        dist.barrier()
    t2 = time.time()
    merged = shard.all_gather_rows(rows, dist, device, capacity=capacity)
    if timings is not None:                 # the phases of the N > 1 leg, per rank: the bundle (this rank's shard), the wait, the one collective
        timings["bundle_s"] = timings.get("bundle_s", 0.0) + t1 - t0
        timings["wait_for_slowest_rank_s"] = timings.get("wait_for_slowest_rank_s", 0.0) + t2 - t1
        timings["allgather_s"] = timings.get("allgather_s", 0.0) + time.time() - t2
    return merged, res


def write_outputs(res, loci, out_dir, sample_name, ref_fasta, sv_info=None, today=None):
    """Final artefacts of a run (`<sample>.telr.{json,expanded.json,te.fasta,contig.fasta,vcf,bed}`) from the in-memory
    results of `run_loci`, through the mirror of the reference's writer (telr_output.generate_output; the reference
    passes the same things as files between stages, TELR_output.py:10-20).

    loci: the dicts given to run_loci (name "<chr>_<start>_<end>", contig);  sv_info: {locus name: (genotype, ref_count,
    alt_count)} from the SV caller's table (columns 10-12 of `<sample>.vcf_filtered.tsv`), default ("./.", "0", number of
    window reads);  ref_fasta: path of the reference FASTA (its .fai gives the ##contig lines, written when absent).
    Returns (final_report, final_report_expanded)."""
    import os
    from . import telr_output
    inter = os.path.join(out_dir, "intermediate_files")
    os.makedirs(inter, exist_ok=True)
    contigs = {l["name"]: l["contig"] for l in loci}
    contig_fa = os.path.join(inter, sample_name + ".contigs.fa")
    with open(contig_fa, "w") as fh:                      # header carries len= as the reference's merged contig file does
        for l in loci:
            n_reads = len(l.get("read_idx", l.get("reads", [])))
            fh.write(">%s len=%d reads=%d\n%s\n" % (l["name"], len(l["contig"]), n_reads, l["contig"]))
    ann_bed = os.path.join(inter, sample_name + ".te2contig_filtered.bed")
    te_fa = os.path.join(inter, sample_name + ".te.fa")
    with open(ann_bed, "w") as fb, open(te_fa, "w") as ft:
        for r in res["annotation"]:
            fb.write("\t".join(str(x) for x in r[:6]) + "\n")
            s, e = int(r[1]), int(r[2])
            ft.write(">%s:%d-%d\n%s\n" % (r[0], s, e, contigs[r[0]][s:e]))
    vcf_parsed = os.path.join(inter, sample_name + ".vcf_filtered.tsv")
    with open(vcf_parsed, "w") as fh:
        for l in loci:
            chrom, start, end = l["name"].rsplit("_", 2)
            n_reads = len(l.get("read_idx", l.get("reads", [])))
            gt, dr, dv = (sv_info or {}).get(l["name"], ("./.", "0", str(n_reads)))
            row = [chrom, start, end, str(len(l.get("alt", ""))), str(n_reads), "NA", l["name"], l.get("alt", ""), "NA", "PASS", gt, str(dr), str(dv), "NA"]
            fh.write("\t".join(row) + "\n")
    return telr_output.generate_output(res["liftover"], res["af"], te_fa, vcf_parsed, ann_bed, contig_fa, out_dir, sample_name, ref_fasta, today=today)
