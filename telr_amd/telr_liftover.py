"""Flank -> reference liftover of contig TE annotations (stage 3 of the path).

Restates the behaviour of the reference's `src/telr/TELR_liftover.py`
(`run_liftover_single_annotation` :393-937, `get_coord` :269-279, `check_nearby_ref` :288-340,
`paf_to_bed` :215-245, `get_paf_info` :356-380, `liftover` :976-1221) on in-memory records:
the two `minimap2 -cx asm10 -N 10 REF flank.fa` subprocesses per locus (:248-266) become ONE
batched engine call for all flanks of all loci against the resident reference index, and the
bedtools calls become `telr_amd.intervals`.  The reference's quirks are kept on purpose and
are pinned by tests/golden/liftover_*.json (captured from the reference's own code):
  * the 5' flank is [start-flank_len+1, start) (one base short), the 3' flank [end, end+flank_len);
    a flank that leaves the contig is dropped (:433-457, :197-198);
  * `get_coord` is called with the 5'/3' arguments swapped (:269 vs :555-557), so on the minus
    strand the gap sign is inverted;
  * 5' hits are filtered to the locus chromosome, 3' hits are not (:479 vs :494);
  * single-flank rescue writes its QC numbers under the `5p_*` keys even for a 3' flank (:894-904);
  * overlapping non-reference calls keep the entry whose te_length STRING is largest (:1128-1129).
"""
import json
import os

from . import intervals as iv


# --------------------------------------------------------------------------------------
# small helpers (reference :269-279, :343-353, :940-952)
def get_coord(start_3p, end_3p, start_5p, end_5p, strand):
    """NB: callers pass (start_5p, end_5p, start_3p, end_3p) — the reference's own argument swap."""
    if strand == "+":
        start, end = end_3p, start_5p
    else:
        start, end = start_3p, end_5p
    gap = end - start
    if start > end:
        start, end = end, start
    return start, end, gap


def absmin(a, b):
    return a if min(abs(a), abs(b)) == abs(a) else b


def choose_new_size(size_ref, size_old, size_new):
    return size_ref - size_old > size_ref - size_new


def check_nums_similar(num1, num2):
    return abs(num1 - num2) / num2 <= 0.1


def flank_intervals(start, end, flank_len, contig_len):
    """-> ((s5, e5) or None, (s3, e3) or None), 0-based half-open on the contig."""
    s5, e5 = int(start) - flank_len + 1, int(start)
    s3, e3 = int(end), int(end) + flank_len
    f5 = None if (e5 > contig_len or s5 < 0) else (s5, e5)
    f3 = None if (e3 > contig_len or s3 < 0) else (s3, e3)
    return f5, f3


def locus_chrom(contig_name, telr_mode=True, different_contig_name=False):
    """chromosome a 5' hit must be on: contigs are named <chr>_<start>_<end> (:461-467)"""
    if different_contig_name:
        return None
    return "_".join(contig_name.split("_")[:-2]) if telr_mode else contig_name


class PafHit(object):
    """the PAF columns the liftover reads: 0,1,4,5,7,8,9,10,11"""
    __slots__ = ("qname", "qlen", "strand", "tname", "ts", "te", "nmatch", "blen", "mapq")

    def __init__(self, qname, qlen, strand, tname, ts, te, nmatch, blen, mapq):
        self.qname, self.qlen, self.strand, self.tname = qname, int(qlen), strand, tname
        self.ts, self.te, self.nmatch, self.blen, self.mapq = int(ts), int(te), int(nmatch), int(blen), int(mapq)


def paf_info(hits):
    """get_paf_info (:356-380): later hits with the same id overwrite earlier ones"""
    d = {}
    for h in hits:
        d["_".join([h.qname, h.tname, str(h.ts), str(h.te)])] = {
            "query_length": h.qlen, "query_mapp_qual": h.mapq, "num_residue_matches": h.nmatch,
            "alignment_block_length": h.blen, "sequence_identity": float(h.nmatch / h.blen)}
    return d


def paf_to_bed(hits, filter_chrom=None):
    """paf_to_bed (:215-245) -> sorted 6-column BED rows (strings)"""
    rows = [[h.tname, str(h.ts), str(h.te), h.qname, str(h.mapq), h.strand] for h in hits
            if filter_chrom is None or h.tname == filter_chrom]
    return iv.bed_sort(rows)


def check_nearby_ref(chrom, start_query, end_query, family, strand, ref_rows, threshold=5000):
    """signed distance to the nearest same-family same-strand reference TE within `threshold` (:288-340)"""
    distance = None
    if ref_rows:
        q = [[chrom, str(start_query), str(end_query), family, ".", strand]]
        for e in iv.closest_signed_k(q, ref_rows, k=5):
            if e[0] == e[6] and e[3] == e[9] and e[5] == e[11]:
                d_new = int(e[12])
                distance = d_new if distance is None else absmin(distance, d_new)
    if distance is not None and abs(distance) > threshold:
        distance = None
    return distance


_UNLIFTED_KEYS = ["chrom", "start", "end", "strand", "gap", "TSD_length", "TSD_sequence", "5p_flank_align_coord",
                  "5p_flank_mapping_quality", "5p_flank_num_residue_matches", "5p_flank_alignment_block_length",
                  "5p_flank_sequence_identity", "3p_flank_align_coord", "3p_flank_mapping_quality",
                  "3p_flank_num_residue_matches", "3p_flank_alignment_block_length", "3p_flank_sequence_identity",
                  "distance_5p_flank_ref_te", "distance_3p_flank_ref_te"]


def lift_annotation(chrom, start, end, family, strand, hits5, hits3, ref_rows, ref_seq, flank_len=500,
                    flank_gap_max=20, flank_overlap_max=20, different_contig_name=False, telr_mode=True):
    """One annotation -> the reference's per-annotation report dict (`lift_entries`).

    hits5 / hits3: list of PafHit for the 5' / 3' flank, or None when that flank could not be cut.
    ref_rows: reference TE annotation as BED rows (>= 6 string columns) or None.
    ref_seq(chrom) -> reference sequence string (for the TSD sequence).
    """
    prefix = "_".join([chrom, str(start), str(end)]).replace("|", "_")
    out = {"ID": prefix, "genome1_coord": chrom + ":" + str(start) + "-" + str(end)}
    te_length = int(end) - int(start)
    out["te_length"] = te_length
    filter_chrom = locus_chrom(chrom, telr_mode, different_contig_name)

    bed5 = bed3 = None
    qc5, qc3 = {}, {}
    if hits5 is not None:
        qc5 = paf_info(hits5)
        bed5 = paf_to_bed(hits5, filter_chrom)
    if hits3 is not None:
        qc3 = paf_info(hits3)
        bed3 = paf_to_bed(hits3, None)

    reports, num_hits, reported = [], 0, False
    lift_start = lift_end = 0
    if bed5 and bed3:
        for e in iv.closest_same_strand(bed5, bed3):
            if not (e[0] == e[6] and e[6] != "."):
                continue
            lift_chrom, flank_strand = e[0], e[5]
            q5 = qc5["_".join([e[3], e[0], e[1], e[2]])]
            q3 = qc3["_".join([e[9], e[6], e[7], e[8]])]
            s5, e5, s3, e3 = int(e[1]), int(e[2]), int(e[7]), int(e[8])
            lift_start, lift_end, gap = get_coord(s5, e5, s3, e3, flank_strand)
            lift_strand = "+" if flank_strand == strand else "-"
            entry = {
                "type": None, "family": family, "chrom": lift_chrom, "start": int(lift_start), "end": int(lift_end),
                "strand": lift_strand, "gap": gap, "TSD_length": None, "TSD_sequence": None,
                "5p_flank_align_coord": "%s:%d-%d" % (e[0], s5, e5), "5p_flank_mapping_quality": int(e[4]),
                "5p_flank_num_residue_matches": q5["num_residue_matches"],
                "5p_flank_alignment_block_length": q5["alignment_block_length"],
                "5p_flank_sequence_identity": q5["sequence_identity"],
                "3p_flank_align_coord": "%s:%d-%d" % (e[6], s3, e3), "3p_flank_mapping_quality": int(e[10]),
                "3p_flank_num_residue_matches": q3["num_residue_matches"],
                "3p_flank_alignment_block_length": q3["alignment_block_length"],
                "3p_flank_sequence_identity": q3["sequence_identity"],
                "distance_5p_flank_ref_te": None, "distance_3p_flank_ref_te": None, "comment": None,
            }
            d5 = check_nearby_ref(lift_chrom, s5, e5, family, lift_strand, ref_rows)
            d3 = check_nearby_ref(lift_chrom, s3, e3, family, lift_strand, ref_rows)
            if d5 is not None:
                entry["distance_5p_flank_ref_te"] = d5
            if d3 is not None:
                entry["distance_3p_flank_ref_te"] = d3
            te_between = (d5 is not None and 0 <= d5 <= gap and d3 is not None and d3 <= 0 and -d3 <= gap)
            if gap < -flank_overlap_max:
                continue
            if gap <= flank_gap_max:
                if te_between or check_nums_similar(gap, te_length) or gap >= te_length:
                    entry["type"] = "reference"
                    entry["comment"] = "overlap/gap size between 3p and 5p flanks within threshold, include genome2 TE in between"
                else:
                    entry["type"] = "non-reference"
                    entry["comment"] = "overlap/gap size between 3p and 5p flanks within threshold"
                    if gap == 0:
                        entry["TSD_length"] = 0
                    if gap < 0:
                        entry["TSD_length"] = -gap
                        entry["TSD_sequence"] = ref_seq(lift_chrom)[lift_start:lift_end]
                    num_hits += 1
            elif gap <= 0.5 * te_length:
                if te_between:
                    entry["type"] = "reference"
                    entry["comment"] = "flanks gap size less than half of TE annotation, include genome2 TE in between"
                else:
                    entry["type"] = "non-reference"
                    entry["comment"] = "flanks gap size exceeds threshold but less than half of TE annotation, no genome2 TE in between"
                    num_hits += 1
            elif gap <= 20000:          # here gap >= 0.5 * te_length
                entry["type"] = "reference"
                entry["comment"] = ("flanks gap size greater than half of TE annotation, include genome2 TE in between" if te_between
                                    else "flanks gap size greater than half of TE annotation, no genome2 TE in between")
            else:
                continue
            reports.append(entry)
            reported = True

    report = reports
    if len(reports) > 1:
        best_ref, best_nonref = None, None
        for r in reports:
            if r["type"] == "reference":
                if best_ref is None or choose_new_size(te_length, best_ref["gap"], r["gap"]):
                    best_ref = r
            if r["type"] == "non-reference":
                if best_nonref is None:
                    best_nonref = r
                else:
                    reported = False     # two non-reference placements: ambiguous
        report = None
        if reported:
            report = best_nonref if best_nonref is not None else best_ref
            if report is None:
                reported = False
    elif len(reports) == 1:
        report = reports[0]

    if not reported:
        entry = {"type": "unlifted", "family": family}
        for k in _UNLIFTED_KEYS:
            entry[k] = None
        entry["comment"] = "flank alignments not nearby each other / only one flank aligned"
        c5 = ["%s:%s-%s" % (r[0], r[1], r[2]) for r in (bed5 or [])]
        c3 = ["%s:%s-%s" % (r[0], r[1], r[2]) for r in (bed3 or [])]
        if len(c5) == 1:
            entry["5p_flank_align_coord"] = c5[0]
        elif len(c5) > 1:
            entry["5p_flank_align_coord"] = c5
        if len(c3) == 1:
            entry["3p_flank_align_coord"] = c3[0]
        elif len(c3) > 1:
            entry["3p_flank_align_coord"] = c3
        single = None
        if len(c5) == 1 and len(c3) == 0:
            single = ("5p", bed5[0], qc5)
        elif len(c5) == 0 and len(c3) == 1:
            single = ("3p", bed3[0], qc3)
        if single is not None:
            side, r, qcs = single
            f_chrom, f_start, f_end, f_mq, f_strand = r[0], int(r[1]), int(r[2]), int(r[4]), r[5]
            qc = qcs["_".join([r[3], r[0], r[1], r[2]])]
            lift_strand = "+" if f_strand == strand else "-"
            if side == "5p":
                pos = f_end if f_strand == "+" else f_start
            else:
                pos = f_start if f_strand == "+" else f_end
            entry["chrom"] = f_chrom
            entry["start"] = int(pos)
            entry["end"] = int(pos)
            entry["mapp_quality_5p"] = f_mq
            entry["strand"] = lift_strand
            entry["5p_flank_num_residue_matches"] = qc["num_residue_matches"]
            entry["5p_flank_alignment_block_length"] = qc["alignment_block_length"]
            entry["5p_flank_sequence_identity"] = qc["sequence_identity"]
            dist = check_nearby_ref(f_chrom, f_start, f_end, family, lift_strand, ref_rows)
            entry["distance_%s_flank_ref_te" % side] = dist
            if dist is not None and abs(dist) <= 5:
                entry["type"] = "reference"
                entry["comment"] = "only one flank aligned, flank alignment adjacent to reference TE"
            else:
                entry["type"] = "non-reference"
                entry["comment"] = "only one flank aligned, flank alignment not adjacent to reference TE"
                num_hits = 1
        report = entry
    out["report"] = report
    out["num_hits"] = num_hits
    return out


def dedup_reports(data):
    """Drop overlapping non-reference calls (reference :1062-1141): entries with num_hits == 1 and a
    non-reference report are merged by overlap; in every merged group only the entry whose te_length,
    compared AS A STRING, is largest (first on ties) survives."""
    rows = []
    for e in data:
        if e["num_hits"] == 1 and e["report"]["type"] == "non-reference":
            r = e["report"]
            rows.append([r["chrom"], str(r["start"]), str(r["end"]), r["family"], ".", r["strand"], str(e["te_length"]), e["ID"]])
    rows = iv.bed_sort(rows)
    remove = set()
    groups = []
    for r in rows:
        s, en = int(r[1]), int(r[2])
        if groups and groups[-1]["chrom"] == r[0] and s <= groups[-1]["end"]:
            groups[-1]["end"] = max(groups[-1]["end"], en)
            groups[-1]["rows"].append(r)
        else:
            groups.append({"chrom": r[0], "end": en, "rows": [r]})
    for g in groups:
        if len(g["rows"]) > 1:
            lens = [r[6] for r in g["rows"]]
            ids = [r[7] for r in g["rows"]]
            keep = ids[lens.index(max(lens))]
            remove.update(i for i in ids if i != keep)
    return [e for e in data if e["ID"] not in remove]


def summarize(data):
    summ = {"non-reference": {"total": 0, "comments": {}}, "reference": {"total": 0, "comments": {}},
            "unlifted": {"total": 0, "comments": {}}}
    for item in data:
        info = item["report"]
        if info["type"] in summ:
            s = summ[info["type"]]
            s["total"] += 1
            if "comment" in info:
                s["comments"][info["comment"]] = s["comments"].get(info["comment"], 0) + 1
    return summ


def nonref_bed_rows(data):
    return [[i["report"]["chrom"], str(i["report"]["start"]), str(i["report"]["end"]), i["report"]["family"], ".",
             i["report"]["strand"]] for i in data if i["num_hits"] == 1]


def hits_from_result(res, qnames, tnames):
    """engine records -> {query index: [PafHit]} in output order"""
    out = {}
    for a in res.alns:
        out.setdefault(int(a["qid"]), []).append(PafHit(
            qnames[a["qid"]], a["qlen"], "-" if a["flags"] & 8 else "+", tnames[a["tid"]], a["ts"], a["te"],
            a["mlen"], a["blen"], a["mapq"]))
    return out


def engine_flank_mapper(engine_index, ref_names, map_opt=None):
    """flank mapper backed by the HIP engine: ONE batched `-cx asm10 -N 10` call for all flanks"""
    from .presets import preset
    if map_opt is None:
        _, map_opt = preset("asm10")
        map_opt.best_n = 10                     # -N 10  (:254-264)

    def mapper(queries, qnames):
        if not queries:
            return {}
        return hits_from_result(engine_index.map(queries, map_opt), qnames, ref_names)
    return mapper


def parse_paf_line(line):
    e = line.rstrip("\n").split("\t")
    return PafHit(e[0], e[1], e[4], e[5], e[7], e[8], e[9], e[10], e[11])


def liftover(mapper, contigs, annotations, ref_seq, ref_rows=None, flank_len=500, flank_gap_max=20,
             flank_overlap_max=20, out_dir=None, different_contig_name=False, telr_mode=True):
    """Batched replacement of `liftover()` (:976-1221).

    mapper(queries, qnames) -> {query index: [PafHit]} (see engine_flank_mapper).
    contigs: {contig name: sequence}; annotations: BED rows (contig, start, end, family, score, strand);
    ref_seq(chrom) -> reference sequence.  Returns (reports after de-duplication, summary) and writes
    liftover_report.json / liftover_nonref.bed / liftover_summary.json into out_dir when given.
    """
    queries, qnames, owner = [], [], []
    for k, a in enumerate(annotations):
        c = contigs[a[0]]
        f5, f3 = flank_intervals(int(a[1]), int(a[2]), flank_len, len(c))
        for side, f in (("5p", f5), ("3p", f3)):
            if f is not None:
                queries.append(c[f[0]:f[1]])
                qnames.append("%s:%d-%d" % (a[0], f[0], f[1]))      # bedtools getfasta header
                owner.append((k, side))
    by_q = mapper(queries, qnames)
    per = {}
    for qi, (k, side) in enumerate(owner):
        per[(k, side)] = by_q.get(qi, [])
    data = []
    for k, a in enumerate(annotations):
        data.append(lift_annotation(a[0], int(a[1]), int(a[2]), a[3], a[5], per.get((k, "5p")), per.get((k, "3p")),
                                    ref_rows, ref_seq, flank_len, flank_gap_max, flank_overlap_max,
                                    different_contig_name, telr_mode))
    data_new = dedup_reports(data)
    summ = summarize(data_new)
    if out_dir:
        with open(os.path.join(out_dir, "liftover_report.json"), "w") as f:
            json.dump(data_new, f, indent=4, sort_keys=False)
        with open(os.path.join(out_dir, "liftover_nonref.bed"), "w") as f:
            for r in nonref_bed_rows(data_new):
                f.write("\t".join(r) + "\n")
        with open(os.path.join(out_dir, "liftover_summary.json"), "w") as f:
            json.dump(summ, f, indent=4, sort_keys=False)
    return data_new, summ
