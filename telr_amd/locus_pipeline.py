"""The per-locus bundle of the path (stages 3-4) as batched engine calls:
   S4 ALT->contig, S5 library->contig, S7 flanks->reference + liftover decision tree, S6 reads->contig
   forward / reverse-complement + depth medians + allele frequency.
A "locus" = (name "<chr>_<start>_<end>", contig sequence, Sniffles ALT sequence, window reads)."""
from . import telr_te, telr_af, telr_liftover


def run_loci(backend, ref_index, ref_names, ref_seq, loci, lib_names, lib_seqs, presets="ont", ref_te_rows=None,
             flank_len=500, gap=20, overlap=20, af_params=(100, 200, 50, 50), read_set=None):
    """loci: list of dicts(name, contig, alt, reads).  With `read_set` (the stage-1 read SeqSet resident on the
    device) a locus gives `read_idx` (indices into it) instead of `reads`.  -> dict(annotation, liftover, summary, af)"""
    names = [l["name"] for l in loci]
    contigs = {l["name"]: l["contig"] for l in loci}
    ann, s2c, t2c = telr_te.annotate_contig(backend, names, [l["contig"] for l in loci], [l["alt"] for l in loci],
                                            lib_names, lib_seqs, presets)
    mapper = telr_liftover.engine_flank_mapper(ref_index, ref_names)
    reports, summary = telr_liftover.liftover(mapper, contigs, ann, ref_seq, ref_te_rows, flank_len, gap, overlap)
    contig_te = {}
    for r in ann:                       # one annotation per contig feeds the AF step (first one wins, as a dict would)
        contig_te.setdefault(r[0], (int(r[1]), int(r[2])))
    if read_set is not None:
        freqs = telr_af.get_af(backend, contigs, contig_te, {l["name"]: l["read_idx"] for l in loci}, presets, *af_params, read_set=read_set)
    else:
        freqs = telr_af.get_af(backend, contigs, contig_te, {l["name"]: l["reads"] for l in loci}, presets, *af_params)
    return {"annotation": ann, "liftover": reports, "summary": summary, "af": freqs}


def run_loci_distributed(backend, ref_index, ref_names, ref_seq, loci, lib_names, lib_seqs, dist=None, device=None, **kw):
    """Stages 3-4 sharded over the ranks of one node: loci are assigned by LPT on their read
    bases, every rank runs the bundle on its shard, and ONE all-gather of fixed-width rows merges the coordinate /
    allele-frequency table (RCCL over xGMI on GPUs; gloo in the CPU tests).  Variable-length payloads (reports,
    sequences) stay on the owning rank.  -> (merged LOCUS_ROW array sorted by locus id, this rank's full results)"""
    from . import shard
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    costs = [sum(len(r) for r in l["reads"]) + len(l["contig"]) for l in loci]
    mine = shard.shard_loci(costs, world)[rank]
    sub = [loci[i] for i in mine]
    res = run_loci(backend, ref_index, ref_names, ref_seq, sub, lib_names, lib_seqs, **kw) if sub else \
        {"annotation": [], "liftover": [], "summary": {}, "af": {}}
    by_name = {"_".join(r["ID"].split("_")[:3]): r for r in res["liftover"]}
    ids, reps, freqs = [], [], []
    for gi, l in zip(mine, sub):
        r = by_name.get(l["name"])
        if r is None:
            continue
        ids.append(gi); reps.append(r); freqs.append(res["af"].get(l["name"]))
    chrom_ids = {n: i for i, n in enumerate(ref_names)}
    fam_ids = {n: i for i, n in enumerate(lib_names)}
    rows = shard.rows_from_reports(ids, reps, freqs, chrom_ids, fam_ids)
    return shard.all_gather_rows(rows, dist, device), res
