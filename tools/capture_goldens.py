#!/usr/bin/env python3
"""Capture golden vectors from the reference's own Python glue (run ONLY in the build container).

The reference (/root/reference) is imported with in-memory stubs for Bio / pysam and with
`subprocess` replaced by a fake that answers the bedtools / minimap2 / samtools calls of the
liftover and AF paths (canned PAF for minimap2; tools/bedtools_bruteforce.py for bedtools).  Only inputs
and outputs are written (JSON under tests/golden/); no reference source is copied.

  python tools/capture_goldens.py          # rewrites tests/golden/*.json
"""
import io
import json
import os
import shutil
import sys
import tempfile
import types
import zlib

if os.environ.get("PYTHONHASHSEED") != "0":            # the reference builds read-id lists from sets: one fixed string hash, so that a re-run rewrites the same files
    os.environ["PYTHONHASHSEED"] = "0"
    os.execv(sys.executable, [sys.executable] + sys.argv)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference/src"
GOLD = os.path.join(ROOT, "tests", "golden")

import bedtools_bruteforce as bt  # noqa: E402   (tools/: written from the bedtools manual, NOT telr_amd.intervals -- that is the thing under test)


def import_reference():
    sys.path.insert(0, REF)
    bio = types.ModuleType("Bio"); seqio = types.ModuleType("Bio.SeqIO"); bio.SeqIO = seqio
    sys.modules["Bio"] = bio; sys.modules["Bio.SeqIO"] = seqio; sys.modules["pysam"] = types.ModuleType("pysam")
    import telr.TELR_liftover as L
    import telr.TELR_te as T
    import telr.TELR_sv as S
    import telr.TELR_utility as U
    return L, T, S, U


# ---------------------------------------------------------------------------------------
class FakeSubprocess(object):
    """Answers the external-tool calls made by TELR_liftover with in-memory data."""

    def __init__(self, seqs, paf_by_flank):
        self.seqs = seqs                  # {fasta path: {name: sequence}}
        self.paf = paf_by_flank           # {"5p": [paf line, ...], "3p": [...]} or {(locus prefix, side): [...]}
        self.PIPE = -1

    def _rows(self, path):
        with open(path) as f:
            return [l.rstrip("\n").split("\t") for l in f if l.strip()]

    def call(self, cmd, stdout=None, shell=False, **kw):
        if shell:
            cmd = cmd.replace('"', "").split()
        tool = os.path.basename(cmd[0])
        out = ""
        if tool == "bedtools" and cmd[1] == "getfasta":
            fa, bed = cmd[cmd.index("-fi") + 1], cmd[cmd.index("-bed") + 1]
            for r in self._rows(bed):
                s = self.seqs[fa][r[0]][int(r[1]):int(r[2])]
                out += ">%s:%s-%s\n%s\n" % (r[0], r[1], r[2], s)
        elif tool == "bedtools" and cmd[1] == "sort":
            out = "".join("\t".join(r) + "\n" for r in bt.sort_bed(self._rows(cmd[cmd.index("-i") + 1])))
        elif tool == "bedtools" and cmd[1] == "closest":
            a = self._rows(cmd[cmd.index("-a") + 1]); b = self._rows(cmd[cmd.index("-b") + 1])
            rows = bt.closest_s_d_tall(a, b) if "-s" in cmd else bt.closest_d_Dref_k(a, b, int(cmd[cmd.index("-k") + 1]))
            out = "".join("\t".join(r) + "\n" for r in rows)
        elif tool == "bedtools" and cmd[1] == "merge":
            rows = self._rows(cmd[cmd.index("-i") + 1])
            cols = [int(c) - 1 for c in cmd[cmd.index("-c") + 1].split(",")]
            groups = []
            for r in rows:
                s, e = int(r[1]), int(r[2])
                if groups and groups[-1][0] == r[0] and s <= groups[-1][2]:
                    groups[-1][2] = max(groups[-1][2], e); groups[-1][3].append(r)
                else:
                    groups.append([r[0], s, e, [r]])
            for c, s, e, rs in groups:
                out += "\t".join([c, str(s), str(e)] + [",".join(r[k] for r in rs) for k in cols]) + "\n"
        elif tool == "minimap2":
            flank_fa = cmd[-1]
            base = os.path.basename(flank_fa)
            side = "5p" if base.endswith("_5p.fa") else "3p"
            key = (base[:-len("_5p.fa")], side)
            lines = self.paf.get(key, self.paf.get(side, []))
            out = "".join(l + "\n" for l in lines)
        elif tool == "samtools":
            return 0
        else:
            raise RuntimeError("unexpected command: %r" % (cmd,))
        if stdout is not None:
            stdout.write(out)
        return 0


def write_fasta_with_fai(path, seqs):
    with open(path, "w") as f, open(path + ".fai", "w") as g:
        off = 0
        for n, s in seqs.items():
            hdr = ">%s\n" % n
            f.write(hdr + s + "\n")
            off += len(hdr)
            g.write("%s\t%d\t%d\t%d\t%d\n" % (n, len(s), off, len(s), len(s) + 1))
            off += len(s) + 1


def paf(qname, qlen, qs, qe, strand, tname, tlen, ts, te, nmatch, blen, mapq):
    return "\t".join(str(x) for x in [qname, qlen, qs, qe, strand, tname, tlen, ts, te, nmatch, blen, mapq, "tp:A:P", "cm:i:40"])


def rnd_seq(n, seed):
    import random
    r = random.Random(seed)
    return "".join(r.choice("ACGT") for _ in range(n))


def liftover_cases():
    """(name, annotation, contig length, {"5p": [...], "3p": [...]}, ref TE bed rows, gap, overlap)"""
    C = "chr2L_33000_33020"          # telr-mode contig name -> locus chromosome chr2L
    TL = 23513712
    cases = []

    def f5(ts, te, strand="+", t="chr2L", mq=60, nm=480, bl=499):
        return paf("%s:%d-%d" % (C, 4501, 5000), 499, 0, 499, strand, t, TL, ts, te, nm, bl, mq)

    def f3(ts, te, strand="+", t="chr2L", mq=60, nm=490, bl=500):
        return paf("%s:%d-%d" % (C, 9600, 10100), 500, 0, 500, strand, t, TL, ts, te, nm, bl, mq)
    A = dict(chrom=C, start=5000, end=9600, family="jockey", strand="+")
    Am = dict(A, strand="-")
    cases.append(("plus_overlap5_tsd", A, 20000, {"5p": [f5(32519, 33018)], "3p": [f3(33013, 33513)]}, None, 20, 20))
    cases.append(("minus_overlap5", A, 20000, {"5p": [f5(33013, 33512, "-")], "3p": [f3(32518, 33018, "-")]}, None, 20, 20))
    cases.append(("minus_gap5", Am, 20000, {"5p": [f5(33023, 33522, "-")], "3p": [f3(32518, 33018, "-")]}, None, 20, 20))
    cases.append(("plus_gap0", A, 20000, {"5p": [f5(32519, 33018)], "3p": [f3(33018, 33518)]}, None, 20, 20))
    cases.append(("plus_gap10", A, 20000, {"5p": [f5(32519, 33018)], "3p": [f3(33028, 33528)]}, None, 20, 20))
    cases.append(("plus_gap_equals_te_reference", A, 20000, {"5p": [f5(32519, 33018)], "3p": [f3(33018 + 4600, 33518 + 4600)]}, None, 20, 20))
    cases.append(("plus_gap_lt_half_te_nonref", A, 20000, {"5p": [f5(32519, 33018)], "3p": [f3(33018 + 900, 33518 + 900)]}, None, 20, 20))
    cases.append(("plus_gap_gt_half_te_reference", A, 20000, {"5p": [f5(32519, 33018)], "3p": [f3(33018 + 3000, 33518 + 3000)]}, None, 20, 20))
    cases.append(("plus_gap_gt_20kb_unlifted", A, 20000, {"5p": [f5(32519, 33018)], "3p": [f3(33018 + 30000, 33518 + 30000)]}, None, 20, 20))
    cases.append(("plus_big_overlap_unlifted", A, 20000, {"5p": [f5(32519, 33018)], "3p": [f3(32900, 33400)]}, None, 20, 20))
    ref_te = [["chr2L", "33100", "33900", "jockey", ".", "+"], ["chr2L", "50000", "51000", "roo", ".", "+"],
              ["chr2L", "33100", "33900", "jockey", ".", "-"]]
    cases.append(("plus_ref_te_between", A, 20000, {"5p": [f5(32519, 33018)], "3p": [f3(33950, 34450)]}, ref_te, 20, 20))
    cases.append(("plus_ref_te_other_family", dict(A, family="roo"), 20000, {"5p": [f5(32519, 33018)], "3p": [f3(33950, 34450)]}, ref_te, 20, 20))
    cases.append(("single_5p_flank", dict(A, end=19800), 20000, {"5p": [f5(32519, 33018)]}, None, 20, 20))
    cases.append(("single_3p_flank_minus", dict(A, start=300), 20000, {"3p": [f3(32518, 33018, "-")]}, None, 20, 20))
    cases.append(("single_5p_adjacent_ref_te", dict(A, end=19800), 20000, {"5p": [f5(32601, 33100)]}, ref_te, 20, 20))
    cases.append(("offchrom_5p_hit_filtered", A, 20000, {"5p": [f5(1000, 1499, "+", "chr3R")], "3p": [f3(33013, 33513)]}, None, 20, 20))
    cases.append(("two_nonref_placements_ambiguous", A, 20000,
                  {"5p": [f5(32519, 33018), f5(72519, 73018, "+", "chr2L", 3)], "3p": [f3(33013, 33513), f3(73013, 73513, "+", "chr2L", 2)]}, None, 20, 20))
    cases.append(("ref_and_nonref_pick_nonref", A, 20000,
                  {"5p": [f5(32519, 33018), f5(72519, 73018, "+", "chr2L", 3)], "3p": [f3(33013, 33513), f3(73018 + 4600, 73518 + 4600, "+", "chr2L", 2)]}, None, 20, 20))
    cases.append(("two_refs_pick_by_gap", A, 20000,
                  {"5p": [f5(32519, 33018), f5(72519, 73018, "+", "chr2L", 3)], "3p": [f3(33018 + 4000, 33518 + 4000), f3(73018 + 4600, 73518 + 4600, "+", "chr2L", 2)]}, None, 20, 20))
    cases.append(("no_hits", A, 20000, {"5p": [], "3p": []}, None, 20, 20))
    cases.append(("opposite_strand_hits_unpaired", A, 20000, {"5p": [f5(32519, 33018, "+")], "3p": [f3(33013, 33513, "-")]}, None, 20, 20))
    cases.append(("gap50_threshold50", A, 20000, {"5p": [f5(32519, 33018)], "3p": [f3(33058, 33558)]}, None, 50, 50))
    cases.append(("tie_two_3p_hits_same_distance", A, 20000, {"5p": [f5(32519, 33018)], "3p": [f3(33013, 33513), f3(33013, 33400, "+", "chr2L", 10)]}, None, 20, 20))
    return cases


def capture_liftover(L):
    ref_seqs = {"chr2L": rnd_seq(120000, 7), "chr3R": rnd_seq(5000, 8)}
    out_cases = []
    for name, ann, clen, pafs, ref_te, gap, overlap in liftover_cases():
        tmp = tempfile.mkdtemp(prefix="gold_")
        try:
            contigs = {ann["chrom"]: rnd_seq(clen, 11)}
            fa1, fa2 = os.path.join(tmp, "contigs.fa"), os.path.join(tmp, "ref.fa")
            write_fasta_with_fai(fa1, contigs); write_fasta_with_fai(fa2, ref_seqs)
            bed2 = None
            if ref_te:
                bed2 = os.path.join(tmp, "ref_te.bed")
                with open(bed2, "w") as f:
                    for r in ref_te:
                        f.write("\t".join(r) + "\n")
            odir = os.path.join(tmp, "out"); os.mkdir(odir)
            inp = dict(ann, fasta1=fa1, fasta2=fa2, out_dir=odir, flank_len=500, flank_gap_max=gap, flank_overlap_max=overlap,
                       bed2=bed2, preset="asm10", different_contig_name=False, telr_mode=True)
            ij = os.path.join(tmp, "in.json")
            with open(ij, "w") as f:
                json.dump(inp, f)
            L.subprocess = FakeSubprocess({fa1: contigs, fa2: ref_seqs}, pafs)
            rep = L.run_liftover_single_annotation(ij)
            with open(rep) as f:
                expected = json.load(f)
            out_cases.append({"name": name, "annotation": ann, "contig_length": clen, "paf": pafs, "ref_te_bed": ref_te,
                              "flank_gap_max": gap, "flank_overlap_max": overlap, "expected": expected})
        finally:
            shutil.rmtree(tmp)
    return {"ref_seed": {"chr2L": [120000, 7], "chr3R": [5000, 8]}, "cases": out_cases}


def capture_liftover_driver(L):
    """whole liftover(): fan-out over annotations + overlap de-dup + reports (reference :976-1221)"""
    ref_seqs = {"chr2L": rnd_seq(120000, 7)}
    TL = 120000
    contigs, bed1, pafs = {}, [], {}
    # five loci; 2 and 3 lift to overlapping places (te_length "999" vs "1000": string max keeps "999")
    spec = [("chr2L_10000_10010", 5000, 6000, "roo", "+", 10000), ("chr2L_20000_20010", 5000, 5999, "jockey", "+", 20000),
            ("chr2L_20003_20013", 5000, 6000, "jockey", "+", 20003), ("chr2L_40000_40010", 5000, 7000, "copia", "-", 40000),
            ("chr2L_60000_60010", 300, 900, "roo", "+", 60000)]
    for name, s, e, fam, strand, pos in spec:
        contigs[name] = rnd_seq(12000, zlib.crc32(name.encode()) % 1000)
        bed1.append([name, str(s), str(e), fam, ".", strand])
        prefix = "_".join([name, str(s), str(e)])
        q5 = "%s:%d-%d" % (name, s - 499, s); q3 = "%s:%d-%d" % (name, e, e + 500)
        if s - 499 >= 0:
            pafs[(prefix, "5p")] = [paf(q5, 499, 0, 499, "+", "chr2L", TL, pos - 499, pos, 490, 499, 60)]
        pafs[(prefix, "3p")] = [paf(q3, 500, 0, 500, "+", "chr2L", TL, pos - 6, pos + 494, 495, 500, 60)]
    tmp = tempfile.mkdtemp(prefix="gold_")
    try:
        fa1, fa2 = os.path.join(tmp, "contigs.fa"), os.path.join(tmp, "ref.fa")
        write_fasta_with_fai(fa1, contigs); write_fasta_with_fai(fa2, ref_seqs)
        b1 = os.path.join(tmp, "te.bed")
        with open(b1, "w") as f:
            for r in bed1:
                f.write("\t".join(r) + "\n")
        odir = os.path.join(tmp, "out"); os.mkdir(odir)
        L.subprocess = FakeSubprocess({fa1: contigs, fa2: ref_seqs}, pafs)

        class FakePool(object):
            def __init__(self, processes=None):
                pass

            def map(self, fn, items):
                return [fn(i) for i in items]

            def close(self):
                pass

            def join(self):
                pass
        L.Pool = FakePool
        saved = sys.stdout; sys.stdout = io.StringIO()
        try:
            rep = L.liftover(fa1, fa2, b1, None, "asm10", 500, 20, 20, odir, 1, False, False, True)
        finally:
            sys.stdout = saved
        with open(rep) as f:
            report = json.load(f)
        with open(os.path.join(odir, "liftover_nonref.bed")) as f:
            nonref = f.read()
        with open(os.path.join(odir, "liftover_summary.json")) as f:
            summ = json.load(f)
    finally:
        shutil.rmtree(tmp)
    return {"ref_seed": {"chr2L": [120000, 7]}, "contig_seeds": {n: [12000, zlib.crc32(n.encode()) % 1000] for n in contigs},
            "contig_seqs": contigs, "bed1": bed1, "paf": {"|".join(k): v for k, v in pafs.items()},
            "expected_report": report, "expected_nonref_bed": nonref, "expected_summary": summ}


# ---------------------------------------------------------------------------------------
def capture_af(T):
    """get_af (TELR_te.py:578-838) with the aligner / samtools side replaced by a median lookup table."""
    cases_in = [
        # name, te (start,end), contig length, medians fw (te5, te3, fl5, fl3), medians rc
        ("concordant", (3000, 7600), 12000, (13, 12, 18, 17), (14, 15, 18, 19)),
        ("discordant", (3000, 7600), 12000, (4, 4, 18, 17), (16, 15, 18, 19)),
        ("ratio_gt_1p5_one_side", (3000, 7600), 12000, (30, 30, 18, 17), (14, 15, 18, 19)),
        ("cap_at_one", (3000, 7600), 12000, (20, 19, 18, 17), (21, 15, 18, 19)),
        ("zero_te_cov", (3000, 7600), 12000, (0, 0, 18, 17), (0, 0, 18, 19)),
        ("zero_flank_cov", (3000, 7600), 12000, (10, 10, 0, 17), (9, 9, 0, 19)),
        ("short_te_whole_locus", (3000, 3140), 12000, (13, 13, 18, 17), (14, 14, 18, 19)),
        ("te_near_contig_start", (250, 4000), 12000, (13, 12, 18, 17), (14, 15, 18, 19)),
        ("te_near_contig_end", (3000, 11800), 12000, (13, 12, 18, 17), (14, 15, 18, 19)),
        ("half_medians", (3000, 7600), 12000, (12.5, 12, 18, 17.5), (13.5, 15, 18, 19)),
    ]
    out = []
    for name, (s, e), clen, mfw, mrc in cases_in:
        tmp = tempfile.mkdtemp(prefix="gold_")
        try:
            locus = "chr2L_33000_33020"
            cdir = os.path.join(tmp, "contigs"); os.mkdir(cdir)
            open(os.path.join(cdir, locus + ".cns.ctg1.fa"), "w").write(">x\nA\n")
            ann = os.path.join(tmp, "te.bed")
            open(ann, "w").write("\t".join([locus, str(s), str(e), "jockey", ".", "+"]) + "\n")
            vcf = os.path.join(tmp, "vcf.tsv")
            open(vcf, "w").write("\t".join(["chr2L", "33000", "33020", "4600", "12", "0.7", "id1", "ACGT", "r1,r2", "PASS", "0/1", "5", "12", "0.9"]) + "\n")
            requested = []

            def fake_prep(vcf_parsed, o, sample, bam, reads, rdir, read_type="sv"):
                os.makedirs(rdir, exist_ok=True)
                open(os.path.join(rdir, "contig0"), "w").write(">r\nA\n")

            class FakePool(object):
                def __init__(self, processes=None):
                    pass

                def map(self, fn, items):
                    for it in items:
                        open(it[3] + ".realign.sort.bam", "w").write("x")

                def close(self):
                    pass

                def join(self):
                    pass

            def fake_rc(a, b):
                open(b, "w").write(">x\nT\n")

            def fake_len(c):
                return clen

            def fake_median(bam, chrom, start, end):
                rc = ".revcomp." in os.path.basename(bam)
                requested.append(["rc" if rc else "fw", start, end])
                te_s, te_e = (clen - e, clen - s) if rc else (s, e)
                m = mrc if rc else mfw
                # which of the four intervals is it?
                ivs = []
                if te_s + 50 + 50 < te_e:
                    ivs = [(te_s + 50, te_s + 100), (te_e - 100, te_e - 50)]
                else:
                    ivs = [(te_s, te_e), (te_s, te_e)]
                ivs += [(te_s - 300, te_s - 200), (te_e + 200, te_e + 300)]
                for k, x in enumerate(ivs):
                    if (start, end) == x:
                        return m[k]
                raise RuntimeError("unexpected interval %r" % ((start, end),))
            T.prep_assembly_inputs = fake_prep; T.Pool = FakePool; T.get_rev_comp_sequence = fake_rc
            T.get_contig_length = fake_len; T.get_median_cov = fake_median
            freq = T.get_af(tmp, "s", "bam", "reads", ann, cdir, vcf, 100, 200, 50, 50, "ont", 1)
            out.append({"name": name, "te": [s, e], "contig_length": clen, "medians_fw": list(mfw), "medians_rc": list(mrc),
                        "params": [100, 200, 50, 50], "requested_intervals": requested, "expected": freq[locus]})
        finally:
            shutil.rmtree(tmp)
    return {"cases": out}


def capture_helpers(L, T, S, U):
    g = {}
    g["get_coord"] = [[list(a), list(L.get_coord(*a))] for a in
                      [(100, 600, 595, 1095, "+"), (100, 600, 595, 1095, "-"), (595, 1095, 100, 600, "-"), (100, 600, 600, 1100, "+"), (100, 600, 700, 1200, "-")]]
    g["absmin"] = [[[a, b], L.absmin(a, b)] for a, b in [(3, -2), (-3, 2), (-2, 2), (2, -2), (0, 5), (-7, -1)]]
    g["choose_new_size"] = [[[a, b, c], L.choose_new_size(a, b, c)] for a, b, c in [(4600, 4000, 4500), (4600, 4500, 4000), (4600, 5000, 4700), (100, 10, 10)]]
    g["check_nums_similar"] = [[[a, b], L.check_nums_similar(a, b)] for a, b in [(4600, 4600), (4140, 4600), (4139, 4600), (5060, 4600), (5061, 4600), (-5, 100)]]
    g["get_te_flank_ratio"] = [[[a, b], T.get_te_flank_ratio(a, b)] for a, b in [(13.0, 18.0), (0.0, 18.0), (13.0, 0.0), (None, 18.0), (27.0, 18.0), (27.1, 18.0), (18.0, 18.0)]]
    g["format_time"] = [[t, U.format_time(t)] for t in [0.4, 59.6, 61, 3599, 3600, 86399, 90061]]
    g["average"] = [[l, S.average(l)] for l in ["10;11", "1;2", "3", "10;11;12;13", "0;1", "20;21"]]
    return g


def capture_sv(S, U):
    """merge_vcf (TELR_sv.py:84-140): the `bedtools merge -o collapse -c 2..14 -delim ";" -d 20` call is replaced by a
    hand-made intermediate (bedtools is not in this image; rows written from its documented behaviour), the reference's
    own post-processing of that intermediate is what is captured.  Also af_sum / id_merge (as sets) / create_loci_set."""
    def row(c, s, e, ln, cov, af, sid, seq, reads, flt, gt, dr, dv, prop):
        return [c, str(s), str(e), str(ln), str(cov), str(af), sid, seq, reads, flt, gt, str(dr), str(dv), str(prop)]
    table = [
        row("chr2L", 1000, 1002, 999, 10, 0.4, "7", "ACGTAC", "r1,r2,r3", "PASS", "0/1", 12, 3, 0.91),
        row("chr2L", 1015, 1021, 1000, 11, 0.5, "8", "ACGTACGG", "r3,r4", "PASS", "0/1", 11, 2, 0.95),
        row("chr2L", 1041, 1042, 85, 3, 0.3, "9", "AC", "r9", "PASS", "1/1", 2, 1, 0.5),
        row("chr2L", 5000, 5001, 300, 7, 0.6, "10", "ACG", "r5,r6", "PASS", "0/1", 6, 2, 0.8),
        row("chr3R", 10, 11, 450, 4, 0.7, "11", "ACGT", "r7", "PASS", "1/1", 0, 1, 0.99),
        row("chr3R", 31, 32, 460, 5, 0.2, "12", "ACGTT", "r8,r7", "PASS", "0/1", 4, 2, 0.97),
    ]
    # groups under -d 20: {0,1,2} (1002->1015 gap 13, 1021->1041 gap 20), {3}, {4,5} (11->31 gap 20)
    def merged(rows):
        cols = [";".join(r[k] for r in rows) for k in range(1, 14)]
        return [rows[0][0], str(min(int(r[1]) for r in rows)), str(max(int(r[2]) for r in rows))] + cols
    inter = [merged(table[0:3]), merged(table[3:4]), merged(table[4:6])]
    tmp = tempfile.mkdtemp(prefix="gold_")
    try:
        vin = os.path.join(tmp, "in.tsv")
        open(vin, "w").write("".join("\t".join(r) + "\n" for r in table))

        def fake_call(command, shell=False, stdout=None):
            assert "bedtools merge" in command and "-d 20" in command
            stdout.write("".join("\t".join(r) + "\n" for r in inter))
            return 0
        S.subprocess = types.SimpleNamespace(call=fake_call)
        vout = os.path.join(tmp, "out.tsv")
        S.merge_vcf(vin, vout)
        out = [l.split("\t") for l in open(vout).read().splitlines()]
        loci = sorted(U.create_loci_set(vout))
    finally:
        shutil.rmtree(tmp)
    extra = capture_sv_table(S, U)
    return {"table": table, "bedtools_merge": inter, "merged": out, "loci": loci, **extra,
            "af_sum": [[v, S.af_sum(list(v))] for v in ([0.4, 0.5], [0.6, 0.5], [1.0], [0.2, 0.3, 0.6])],
            "id_merge": [[v, sorted(S.id_merge(v).split(","))] for v in (["a,b", "b,c"], ["x"], ["r1,r1", "r1"])]}


def capture_sv_table(S, U):
    """swap_coordinate (TELR_sv.py:183-190), rm_vcf_redundancy (:193-228, pandas groupby) and the table side of filter_vcf
    (:231-324): RepeatMasker is replaced by a canned GFF, `bedtools sort` / `bedtools merge` on that GFF by inline stand-ins
    semantics (GFF is 1-based inclusive; merge prints 0-based starts and joins book-ended features)."""
    parsed = [   # 13 columns, as `bcftools query` prints them (leading blanks in the per-sample fields)
        ["chr2L", "1200", "1100", "310", "9", "0.45", "3", "ACGTACGTAAACGTACGTAA", "r1,r2", "PASS", " 0/1", " 11", " 9"],
        ["chr2L", "1100", "1200", "305", "4", "0.7", "4", "ACGTACGTAAACGTACGTAAACGT", "r2,r5", "PASS", " 0/1", " 3", " 4"],
        ["chr2L", "90", "91", "77", "2", "0.2", "1", "ACGTACGTTT", "r7", "PASS", " 0/0", " 8", " 2"],
        ["chrX", "500", "500", "1000", "6", "0.3", "9", "ACGTACGTAAACGTACGTAAACGTACGTAAACGTACGTAA", "r8,r9,r8", "PASS", " 1/1", " 0", " 6"],
        ["chrX", "500", "500", "990", "5", "0.3", "10", "ACGT", "r9,r10", "PASS", " 1/1", " 1", " 5"],
        ["chr3R", "7", "8", "120", "3", "1.0", "12", "ACGTACGTAAACGTACGTAAACGTACGTAAAC", "r11", "PASS", " 1/1", " 0", " 3"],
    ]
    gff = [  # seqid, start (1-based), end, strand, family
        ("chr2L_1100_1200", 1, 5, "+", "jockey"), ("chr2L_1100_1200", 4, 8, "-", "roo"), ("chr2L_1100_1200", 9, 12, "+", "roo"),
        ("chr2L_1100_1200", 15, 16, "+", "copia"),
        ("chrX_500_500", 3, 6, "+", "jockey"), ("chrX_500_500", 11, 18, "+", "jockey"), ("chrX_500_500", 30, 40, "-", "gypsy"),
        ("chr3R_7_8", 2, 32, "+", "412"),
    ]
    tmp = tempfile.mkdtemp(prefix="gold_")
    try:
        raw, swp, dedup = (os.path.join(tmp, n) for n in ("raw.tsv", "swap.tsv", "dedup.tsv"))
        open(raw, "w").write("".join("\t".join(r) + "\n" for r in parsed))
        S.swap_coordinate(raw, swp)
        swapped = [l.split("\t") for l in open(swp).read().splitlines()]
        S.rm_vcf_redundancy(swp, dedup)
        dedup_text = open(dedup).read()

        merged_bed = []

        def fake_call(cmd, stdout=None, **kw):
            tool = os.path.basename(cmd[0])
            if tool == "RepeatMasker":
                d = cmd[cmd.index("-dir") + 1]
                with open(os.path.join(d, os.path.basename(cmd[-1]) + ".out.gff"), "w") as f:
                    f.write("##gff-version 2\n")
                    for (sid, s, e, st, fam) in gff:
                        f.write("\t".join([sid, "RepeatMasker", "similarity", str(s), str(e), "12.3", st, ".",
                                           'Target "Motif:%s" 1 %d' % (fam, e - s + 1)]) + "\n")
                return 0
            if tool == "bedtools" and cmd[1] == "sort":
                rows = [l.rstrip("\n").split("\t") for l in open(cmd[cmd.index("-i") + 1]) if not l.startswith("#")]
                rows.sort(key=lambda r: (r[0], int(r[3])))
                stdout.write("".join("\t".join(r) + "\n" for r in rows))
                return 0
            if tool == "bedtools" and cmd[1] == "merge":
                rows = [l.rstrip("\n").split("\t") for l in open(cmd[cmd.index("-i") + 1])]
                cur = None
                for r in rows:
                    s0, e = int(r[3]) - 1, int(r[4])
                    if cur is not None and cur[0] == r[0] and s0 <= cur[2]:
                        cur[2] = max(cur[2], e)
                    else:
                        if cur is not None:
                            merged_bed.append(list(cur))
                        cur = [r[0], s0, e]
                if cur is not None:
                    merged_bed.append(list(cur))
                stdout.write("".join("%s\t%d\t%d\n" % tuple(m) for m in merged_bed))
                return 0
            raise AssertionError(cmd)
        S.subprocess = types.SimpleNamespace(call=fake_call)
        S.SeqIO = _fake_seqio()
        out = os.path.join(tmp, "o"); os.mkdir(out)
        filt, ev = os.path.join(tmp, "filt.tsv"), os.path.join(tmp, "eval.tsv")
        open(ev, "w").write("")
        S.filter_vcf(dedup, filt, "lib.fa", out, "s+1", 2, ev)
        filtered_text = open(filt).read()
        eval_rows = sorted(open(ev).read().splitlines())
        ins_fa = open(os.path.join(out, "splus1.vcf_ins.fasta")).read()
    finally:
        shutil.rmtree(tmp)
    return {"parsed": parsed, "swapped": swapped, "dedup_text": dedup_text, "rm_gff": [list(g) for g in gff],
            "rm_merged_bed": merged_bed, "filtered_text": filtered_text, "filter_eval": eval_rows, "ins_fasta": ins_fa}


class _Seq(str):
    def reverse_complement(self):
        return _Seq(self[::-1].translate(str.maketrans("ACGTNacgtn", "TGCANtgcan")))


class _Rec(object):
    def __init__(self, header, seq):
        self.description = header
        self.id = header.split()[0] if header.split() else ""
        self.seq = _Seq(seq)


def _fake_seqio():
    """just enough of Bio.SeqIO for TELR_output.generate_output: parse() and write()"""
    m = types.ModuleType("Bio.SeqIO")

    def parse(handle, fmt):
        hdr, buf = None, []
        for line in handle:
            line = line.rstrip("\n")
            if line.startswith(">"):
                if hdr is not None:
                    yield _Rec(hdr, "".join(buf))
                hdr, buf = line[1:], []
            elif line:
                buf.append(line)
        if hdr is not None:
            yield _Rec(hdr, "".join(buf))

    def write(rec, handle, fmt):
        handle.write(">" + rec.description + "\n")
        s = str(rec.seq)
        for i in range(0, len(s), 60):
            handle.write(s[i:i + 60] + "\n")
    m.parse = parse; m.write = write
    return m


def capture_output(O):
    """generate_output (TELR_output.py:10-297) + write_vcf / write_bed (:300-426) on a hand-made liftover report"""
    import datetime
    O.SeqIO = _fake_seqio()
    O.subprocess = types.SimpleNamespace(call=lambda *a, **k: 0)
    tmp = tempfile.mkdtemp(prefix="gold_")
    try:
        contigs = {"chr2L_33000_33020": rnd_seq(900, 3), "chr2L_50000_50010": rnd_seq(700, 4), "chr3R_100_120": rnd_seq(600, 5), "chrX_5_9": rnd_seq(500, 6)}
        cfa = os.path.join(tmp, "contigs.fa")
        with open(cfa, "w") as f:
            for n, sq in contigs.items():
                f.write(">%s len=%d reads=12\n%s\n" % (n, len(sq), sq))
        ann = os.path.join(tmp, "te.bed")
        open(ann, "w").write("chr2L_33000_33020\t100\t400\tjockey\t.\t-\nchr2L_50000_50010\t50\t300\troo|copia\t.\t.\nchr3R_100_120\t10\t200\tcopia\t.\t+\nchrX_5_9\t20\t90\troo\t.\t+\n")
        tefa = os.path.join(tmp, "te.fa")
        te = {"chr2L_33000_33020:100-400": contigs["chr2L_33000_33020"][100:400], "chr2L_50000_50010:50-300": contigs["chr2L_50000_50010"][50:300],
              "chr3R_100_120:10-200": contigs["chr3R_100_120"][10:200], "chrX_5_9:20-90": contigs["chrX_5_9"][20:90]}
        with open(tefa, "w") as f:
            for n, sq in te.items():
                f.write(">%s\n%s\n" % (n, sq))
        vcf = os.path.join(tmp, "vcf.tsv")
        with open(vcf, "w") as f:
            f.write("\t".join(["chr2L", "33000", "33020", "4600", "12", "0.7", "id1", "ACGT", "r1,r2", "PASS", "0/1", "5", "12", "0.9"]) + "\n")
            f.write("\t".join(["chr2L", "50000", "50010", "250", "9", "1", "id2", "AC GT", "r3", "PASS", "1/1", "0", " 9", "0.8"]) + "\n")
            f.write("\t".join(["chr3R", "100", "120", "190", "4", "0.4", "id3", "AC", "r4", "PASS", "0/1", "7", "4", "0.95"]) + "\n")
            f.write("\t".join(["chrX", "5", "9", "70", "3", "0.3", "id4", "AC", "r5", "PASS", "0/1", "6", "3", "0.99"]) + "\n")

        def rep(t, chrom, s, e, fam, strand, gap, tsd_len, tsd_seq, both=True):
            return {"type": t, "family": fam, "chrom": chrom, "start": s, "end": e, "strand": strand, "gap": gap, "TSD_length": tsd_len, "TSD_sequence": tsd_seq,
                    "5p_flank_align_coord": "%s:%d-%d" % (chrom, s - 499, s) if chrom else None, "5p_flank_mapping_quality": 60, "5p_flank_num_residue_matches": 480,
                    "5p_flank_alignment_block_length": 499, "5p_flank_sequence_identity": 480 / 499,
                    "3p_flank_align_coord": ("%s:%d-%d" % (chrom, e - 5, e + 495)) if both and chrom else None, "3p_flank_mapping_quality": 60 if both else None,
                    "3p_flank_num_residue_matches": 490 if both else None, "3p_flank_alignment_block_length": 500 if both else None,
                    "3p_flank_sequence_identity": 0.98 if both else None, "distance_5p_flank_ref_te": None, "distance_3p_flank_ref_te": None, "comment": "c"}
        lift = [
            {"ID": "chr2L_33000_33020_100_400", "genome1_coord": "chr2L_33000_33020:100-400", "te_length": 300, "num_hits": 1,
             "report": rep("non-reference", "chr2L", 33013, 33018, "jockey", "-", -5, 5, "acgta")},
            {"ID": "chr2L_50000_50010_50_300", "genome1_coord": "chr2L_50000_50010:50-300", "te_length": 250, "num_hits": 1,
             "report": rep("non-reference", "chr2L", 50004, 50004, "roo|copia", "+", None, None, None, both=False)},
            {"ID": "chr3R_100_120_10_200", "genome1_coord": "chr3R_100_120:10-200", "te_length": 190, "num_hits": 0,
             "report": rep("reference", "chr3R", 100, 4700, "copia", "+", 4600, None, None)},
            {"ID": "chrX_5_9_20_90", "genome1_coord": "chrX_5_9:20-90", "te_length": 70, "num_hits": 1,
             "report": rep("non-reference", "chrX", 7, 7, "roo", "+", 0, 0, None)},
        ]
        lj = os.path.join(tmp, "lift.json"); json.dump(lift, open(lj, "w"))
        freq = {"chr2L_33000_33020": {"te_5p_cov": 13.0, "te_3p_cov": 12.0, "flank_5p_cov": 18.0, "flank_3p_cov": 17.0, "te_5p_cov_rc": 14.0, "te_3p_cov_rc": 15.0, "flank_5p_cov_rc": 18.0, "flank_3p_cov_rc": 19.0, "freq": 0.75},
                "chr2L_50000_50010": {"te_5p_cov": 9.0, "te_3p_cov": 9.0, "flank_5p_cov": None, "flank_3p_cov": 8.0, "te_5p_cov_rc": 9.0, "te_3p_cov_rc": 9.0, "flank_5p_cov_rc": 9.0, "flank_3p_cov_rc": None, "freq": 1},
                "chr3R_100_120": {"te_5p_cov": 1.0, "te_3p_cov": 1.0, "flank_5p_cov": 1.0, "flank_3p_cov": 1.0, "te_5p_cov_rc": 1.0, "te_3p_cov_rc": 1.0, "flank_5p_cov_rc": 1.0, "flank_3p_cov_rc": 1.0, "freq": 1},
                "chrX_5_9": {"te_5p_cov": 3.0, "te_3p_cov": 3.0, "flank_5p_cov": 30.0, "flank_3p_cov": 1.0, "te_5p_cov_rc": 0.0, "te_3p_cov_rc": 1.0, "flank_5p_cov_rc": 1.0, "flank_3p_cov_rc": 1.0, "freq": None}}
        ref = os.path.join(tmp, "ref.fa")
        write_fasta_with_fai(ref, {"chr2L": rnd_seq(300, 1), "chr3R": rnd_seq(200, 2), "chrX": "ACGT" * 10})
        odir = os.path.join(tmp, "out"); os.mkdir(odir)
        O.generate_output(lj, freq, tefa, vcf, ann, cfa, odir, "s", ref)
        files = {}
        for n in ("s.telr.json", "s.telr.expanded.json", "s.telr.te.fasta", "s.telr.contig.fasta", "s.telr.vcf", "s.telr.bed"):
            files[n] = open(os.path.join(odir, n)).read().replace(ref, "REF.fa").replace(str(datetime.date.today()), "DATE")
        # empty report
        odir2 = os.path.join(tmp, "out2"); os.mkdir(odir2)
        json.dump([lift[2]], open(lj, "w"))
        O.generate_output(lj, freq, tefa, vcf, ann, cfa, odir2, "e", ref)
        files["e.telr.vcf"] = open(os.path.join(odir2, "e.telr.vcf")).read().replace(ref, "REF.fa").replace(str(datetime.date.today()), "DATE")
        files["e.telr.bed"] = open(os.path.join(odir2, "e.telr.bed")).read()
        files["e.telr.json"] = open(os.path.join(odir2, "e.telr.json")).read()
        # every TSD missing: the columns stay text and print "None"
        odir3 = os.path.join(tmp, "out3"); os.mkdir(odir3)
        json.dump([lift[1]], open(lj, "w"))
        O.generate_output(lj, freq, tefa, vcf, ann, cfa, odir3, "n", ref)
        files["n.telr.vcf"] = open(os.path.join(odir3, "n.telr.vcf")).read().replace(ref, "REF.fa").replace(str(datetime.date.today()), "DATE")
        return {"contigs_fa": open(cfa).read(), "annotation_bed": open(ann).read(), "te_fa": open(tefa).read(), "vcf_parsed": open(vcf).read(),
                "liftover": lift, "te_freq": freq, "ref_fai": open(ref + ".fai").read(), "files": files}
    finally:
        shutil.rmtree(tmp)


def capture_prep_assembly():
    """prep_assembly_inputs(read_type="all") (TELR_assembly.py:384-462) with pysam / seqtk / Bio stubbed: the stage-1 BAM is a
    table of records (read, chromosome, 0-based start, end, SAM flag); `fetch` follows htslib's region rule (a record is
    returned when start < region end and end > region start, secondary and supplementary records included, unmapped
    ones not).  cat | sort | uniq, csplit and cp are the real coreutils."""
    import random
    import subprocess as real_sp
    import telr.TELR_assembly as A
    rnd = random.Random(11)
    chrom_len = {"chr2L": 60000, "chrUn_CP007071v1": 20000, "chr4": 9000}
    recs = []
    names = ["read%03d" % i for i in range(70)]
    for i, n in enumerate(names):
        c = ["chr2L", "chr2L", "chr2L", "chrUn_CP007071v1", "chr4"][i % 5]
        L = rnd.randint(300, 9000); s0 = rnd.randint(0, max(1, chrom_len[c] - L))
        recs.append((n, c, s0, s0 + L, rnd.choice([0, 16])))
        if i % 4 == 0:      # a supplementary piece elsewhere, sometimes on another chromosome
            c2 = c if i % 8 else "chr2L"
            L2 = rnd.randint(200, 3000); s2 = rnd.randint(0, max(1, chrom_len[c2] - L2))
            recs.append((n, c2, s2, s2 + L2, 2048 | rnd.choice([0, 16])))
        if i % 7 == 0:
            L2 = rnd.randint(200, 3000); s2 = rnd.randint(0, max(1, chrom_len[c] - L2))
            recs.append((n, c, s2, s2 + L2, 256))
    # edge cases around the locus at chr2L:30000-30001 (breakpoint 30000, window [29000, 31000))
    recs += [("edge_end_at_start", "chr2L", 28000, 29000, 0),       # ends where the window starts: no overlap
             ("edge_end_past_start", "chr2L", 28000, 29001, 0),     # one base inside
             ("edge_start_at_end", "chr2L", 31000, 33000, 16),      # starts where the window ends: no overlap
             ("edge_start_before_end", "chr2L", 30999, 33000, 16),
             ("edge_secondary_only", "chr2L", 29500, 30500, 256),
             ("edge_unmapped", None, -1, -1, 4)]
    loci = [["chr2L", "30000", "30001"], ["chr2L", "500", "520"], ["chr2L", "10", "11"], ["chr2L", "45001", "45004"], ["chr2L", "45002", "45004"],
            ["chrUn_CP007071v1", "7000", "7001"], ["chr4", "8990", "8991"], ["chr4", "3", "4"]]
    rows = []
    for k, l in enumerate(loci):
        # the 14 columns of <sample>.vcf_filtered.tsv (TELR_sv.py:84-140); column 9 = the SV caller's read list
        rows.append(l + ["100", "5", "fam", "id%d" % k, "ACGT", ",".join(names[k:k + 3]), "PASS", "0/1", "3", "4", "0.5"])

    class Rec(object):
        def __init__(self, n):
            self.query_name = n

    class FakeSam(object):
        def __init__(self, path, mode):
            pass

        def fetch(self, c, start, end):
            if c not in chrom_len:
                raise ValueError("invalid contig `%s`" % c)
            for (n, rc, s0, e0, fl) in recs:
                if rc == c and not (fl & 4) and s0 < end and e0 > start:
                    yield Rec(n)

    class FakeSp(object):
        PIPE = real_sp.PIPE

        def call(self, cmd, stdout=None, shell=False, **kw):
            if shell and cmd.startswith("seqtk subseq"):
                parts = cmd.split()
                fa, ids = parts[2], parts[3]
                want = set(x.strip() for x in open(ids) if x.strip())
                name = None
                for line in open(fa):
                    if line.startswith(">"):
                        name = line[1:].split()[0]
                    if name in want:
                        stdout.write(line)
                return 0
            return real_sp.call(cmd, stdout=stdout, shell=shell, **kw)

    class FakeIndex(dict):
        def get_raw(self, k):
            return self[k]

    def fake_index(path, fmt):
        d, name = FakeIndex(), None
        for line in open(path, "rb"):
            if line.startswith(b">"):
                name = line[1:].split()[0].decode(); d[name] = b""
            d[name] += line
        return d
    tmp = tempfile.mkdtemp()
    try:
        A.pysam.AlignmentFile = FakeSam
        A.subprocess = FakeSp()
        A.SeqIO.index = fake_index
        vcf = os.path.join(tmp, "s.vcf_filtered.tsv")
        with open(vcf, "w") as f:
            for r in rows:
                f.write("\t".join(r) + "\n")
        reads_fa = os.path.join(tmp, "reads.fa")
        with open(reads_fa, "w") as f:
            for n in names + ["edge_end_at_start", "edge_end_past_start", "edge_start_at_end", "edge_start_before_end", "edge_secondary_only", "edge_unmapped"]:
                f.write(">%s\n%s\n" % (n, rnd_seq(40, sum(n.encode()))))
        rdir = os.path.join(tmp, "telr_reads")
        A.prep_assembly_inputs(vcf, tmp, "s", "stage1.bam", reads_fa, rdir, read_type="all")
        new_table = open(vcf + ".new").read()
        per_locus = []
        for k in range(len(loci)):
            ids = [l[1:].split()[0] for l in open(os.path.join(rdir, "contig%d" % k)) if l.startswith(">")]
            per_locus.append(sorted(ids))
        return {"records": [list(r) for r in recs], "chrom_len": chrom_len, "vcf_rows": rows, "expected_new_table": new_table,
                "expected_reads_per_locus": per_locus}
    finally:
        shutil.rmtree(tmp)


def capture_repeatmask(T):
    """parse_rm_out and gff3tobed of the reference on a RepeatMasker .out.gff written here (RepeatMasker's GFF2 lines:
    seq, RepeatMasker, similarity, start, end, score, strand, ., Target "Motif:<family>" start end); `bedtools sort` answered by
    tools/bedtools_bruteforce.sort_bed"""
    import random
    rnd = random.Random(4)
    lines = ["##gff-version 2", "##date 2026-10-02", "##sequence-region ref_38kb.fasta"]
    for k in range(40):
        c = rnd.choice(["chr2L", "chr2R", "chrX", "chr10", "chr4"])
        s = rnd.randint(1, 200000); e = s + rnd.randint(50, 6000)
        lines.append("\t".join([c, "RepeatMasker", "similarity", str(s), str(e), "%.1f" % rnd.uniform(0.1, 30), rnd.choice("+-"), ".",
                                 'Target "Motif:%s" %d %d' % (rnd.choice(["jockey", "roo", "FB4_DM", "I-element", "1360"]), rnd.randint(1, 300), rnd.randint(301, 5000))]))
    d = tempfile.mkdtemp()
    try:
        gff = os.path.join(d, "x.out.gff"); gff3 = os.path.join(d, "x.out.gff3"); bed = os.path.join(d, "x.te.bed")
        with open(gff, "w") as f:
            f.write("\n".join(lines) + "\n")
        T.parse_rm_out(gff, gff3)
        fake = FakeSubprocess({}, {})
        real_call = T.subprocess.call

        def call(cmd, stdout=None, shell=False, **kw):
            toks = cmd.split() if shell else cmd
            assert toks[0] == "bedtools" and toks[1] == "sort"
            rows = fake._rows(toks[toks.index("-i") + 1])
            stdout.write("".join("\t".join(r) + "\n" for r in bt.sort_bed(rows)))
            return 0
        T.subprocess.call = call
        try:
            T.gff3tobed(gff3, bed)
        finally:
            T.subprocess.call = real_call
        return {"rm_out_gff": open(gff).read(), "gff3": open(gff3).read(), "bed": open(bed).read()}
    finally:
        shutil.rmtree(d)


# ---------------------------------------------------------------------------------------
def nested_locus():
    """ONE nested insertion, built from seeds: the reference carries a copy of family famA between two unique flanks; the
    assembled contig carries famB inserted INSIDE that copy (TSD of 6 bases).  -> ref, contig, ALT sequence, library"""
    fl, fr = rnd_seq(3000, 101), rnd_seq(3000, 102)
    fam_a, fam_b = rnd_seq(2400, 103), rnd_seq(1500, 104)
    ref = rnd_seq(20000, 105) + fl + fam_a + fr + rnd_seq(20000, 106)
    cut = 1100
    tsd = fam_a[cut - 6:cut]
    contig = fl + fam_a[:cut] + fam_b + tsd + fam_a[cut:] + fr
    alt = fam_b + tsd                                 # Sniffles' ALT of an insertion: the inserted bases (breakpoint to breakpoint)
    pos = 20000 + len(fl) + cut                      # where famB sits on the reference
    return ref, contig, alt, ["famA", "famB"], [fam_a, fam_b], pos


class FakeAnnot(FakeSubprocess):
    """the tool calls of TELR_te.annotate_contig (minimap2_family=True): samtools faidx, the two minimap2 runs (canned PAF made by
    the CPU oracle from the same sequences), bedtools intersect -wao / sort / merge -d / getfasta (tools/bedtools_bruteforce.py)"""
    def __init__(self, seqs, s4_paf, s5_paf):
        FakeSubprocess.__init__(self, seqs, {})
        self.s4, self.s5 = s4_paf, s5_paf

    def check_output(self, cmd, **kw):                    # TELR_utility.get_cmd_output: the S4 run
        assert os.path.basename(cmd[0]) == "minimap2" and "--secondary=no" in cmd
        return "".join(l + "\n" for l in self.s4).encode()

    def call(self, cmd, stdout=None, shell=False, **kw):
        c = cmd.replace('"', "").split() if shell else list(cmd)
        tool = os.path.basename(c[0])
        if tool == "minimap2":                             # the S5 run: the library against one contig
            stdout.write("".join(l + "\n" for l in self.s5)); return 0
        if tool == "bedtools" and c[1] == "intersect":
            a = self._rows(c[c.index("-a") + 1]); b = self._rows(c[c.index("-b") + 1])
            stdout.write("".join("\t".join(r) + "\n" for r in bt.intersect_wao(a, b))); return 0
        if tool == "bedtools" and c[1] == "merge" and "-d" in c:
            rows = self._rows(c[c.index("-i") + 1])
            cols = [int(x) - 1 for x in c[c.index("-c") + 1].split(",")]
            stdout.write("".join("\t".join(r) + "\n" for r in bt.merge(rows, int(c[c.index("-d") + 1]), cols, ["distinct"] * len(cols), c[c.index("-delim") + 1]))); return 0
        return FakeSubprocess.call(self, cmd, stdout=stdout, shell=shell, **kw)


def capture_nested(T, L, U):
    """The reference's OWN annotate_contig (merge -d 10000 of the library hits that overlap the ALT hit) and its OWN liftover
    on one nested insertion, fed with the aligner output of the CPU oracle (= the engine's, bit for bit) for S4, S5 and S7 --
    run once with the long join in the per-locus presets (bw_long 20000: minimap2 2.22's -r500,20000) and once without.  The
    verdicts are the golden: what the REFERENCE's glue makes of each aligner behaviour."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_backend import OracleBackend
    from telr_amd import presets as P, telr_te, telr_liftover
    ref, contig, alt, lib_names, lib, pos = nested_locus()
    name = "chr2L_%d_%d" % (pos, pos + 1)
    be = OracleBackend()
    out = {"ref_seeds": "tools/capture_goldens.py: nested_locus()", "locus": name, "truth_pos": pos, "cases": {}}
    for label, bwl in (("long_join", 20000), ("no_long_join", 0)):
        with P.override(bw_long=bwl):
            ann, s4rows, s5rows = telr_te.annotate_contig(be, [name], [contig], [alt], lib_names, lib, "ont")
        # PAF text of the two runs (the 12 columns the reference reads: names, lengths, intervals, strand, matches, block, mapq)
        def paf_of(rows, qnames, qlens):
            return [paf(q, ql, 0, ql, r[5], r[0], len(contig), int(r[1]), int(r[2]), int(r[2]) - int(r[1]), int(r[2]) - int(r[1]), int(r[4])) for r, q, ql in zip(rows, qnames, qlens)]
        s4 = paf_of(s4rows, [name] * len(s4rows), [len(alt)] * len(s4rows))
        s5 = paf_of(s5rows, [r[3] for r in s5rows], [len(lib[lib_names.index(r[3])]) for r in s5rows])
        tmp = tempfile.mkdtemp(prefix="gold_")
        try:
            fa1, fa2, lib_fa = os.path.join(tmp, "contigs.fa"), os.path.join(tmp, "ref.fa"), os.path.join(tmp, "lib.fa")
            write_fasta_with_fai(fa1, {name: contig}); write_fasta_with_fai(fa2, {"chr2L": ref}); write_fasta_with_fai(lib_fa, dict(zip(lib_names, lib)))
            vcf = os.path.join(tmp, "vcf.tsv")
            with open(vcf, "w") as f:
                f.write("\t".join(["chr2L", str(pos), str(pos + 1), str(len(alt)), "20", "0.5", "1", alt, "r1,r2", "PASS", "0/1", "10", "10"]) + "\n")
            odir = os.path.join(tmp, "out"); os.mkdir(odir)
            fake = FakeAnnot({fa1: {name: contig}, fa2: {"chr2L": ref}}, s4, s5)
            T.subprocess = fake; U.subprocess = fake
            saved = sys.stdout; sys.stdout = io.StringIO()
            try:
                bed_path, _te_fa = T.annotate_contig(fa1, {name}, lib_fa, vcf, odir, "s", 1, "ont", True, os.path.join(tmp, "eval.tsv"))
            finally:
                sys.stdout = saved
            with open(bed_path) as f:
                ref_ann = [l.rstrip("\n").split("\t") for l in f if l.strip()]
            # S7 for the reference's flanks of that annotation: the oracle maps them (asm10 -N 10), the reference's liftover decides
            io10, mo10 = P.preset("asm10"); mo10.best_n = 10
            rix = be.index([ref], io10)
            pafs = {}
            for r in ref_ann:
                s_, e_ = int(r[1]), int(r[2])
                prefix = "_".join([r[0], r[1], r[2]])
                for side, (a0, a1) in (("5p", (s_ - 499, s_)), ("3p", (e_, e_ + 500))):
                    if a0 < 0 or a1 > len(contig):
                        continue
                    q = contig[a0:a1]
                    res = rix.map([q], mo10)
                    pafs[(prefix, side)] = [paf("%s:%d-%d" % (r[0], a0, a1), len(q), int(a["qs"]), int(a["qe"]), "-" if a["flags"] & 8 else "+", "chr2L", len(ref),
                                                int(a["ts"]), int(a["te"]), int(a["mlen"]), int(a["blen"]), int(a["mapq"])) for a in res.alns]
            b1 = os.path.join(tmp, "te.bed")
            with open(b1, "w") as f:
                for r in ref_ann:
                    f.write("\t".join(r) + "\n")
            # the reference genome's own TE annotation (what RepeatMasker would give): the famA copy
            b2 = os.path.join(tmp, "ref_te.bed")
            with open(b2, "w") as f:
                f.write("\t".join(["chr2L", str(20000 + 3000), str(20000 + 3000 + 2400), "famA", ".", "+"]) + "\n")
            L.subprocess = FakeSubprocess({fa1: {name: contig}, fa2: {"chr2L": ref}}, pafs)

            class FakePool(object):
                def __init__(self, processes=None):
                    pass

                def map(self, fn, items):
                    return [fn(i) for i in items]

                def close(self):
                    pass

                def join(self):
                    pass
            L.Pool = FakePool
            ldir = os.path.join(tmp, "lift"); os.mkdir(ldir)
            saved = sys.stdout; sys.stdout = io.StringIO()
            try:
                rep = L.liftover(fa1, fa2, b1, b2, "asm10", 500, 20, 20, ldir, 1, False, False, True)
            finally:
                sys.stdout = saved
            with open(rep) as f:
                report = json.load(f)
        finally:
            shutil.rmtree(tmp)
        out["cases"][label] = {"bw_long": bwl, "s4_paf": s4, "s5_paf": s5, "reference_annotation": ref_ann, "flank_paf": {"|".join(k): v for k, v in pafs.items()},
                               "reference_liftover_report": report}
    return out


def main():
    L, T, S, U = import_reference()
    if "--only-repeatmask" in sys.argv:
        with open(os.path.join(GOLD, "repeatmask.json"), "w") as f:
            json.dump(capture_repeatmask(T), f, indent=1, sort_keys=True)
        print("wrote repeatmask.json")
        return
    os.makedirs(GOLD, exist_ok=True)
    if "--only-nested" in sys.argv:
        with open(os.path.join(GOLD, "nested_locus.json"), "w") as f:
            json.dump(capture_nested(T, L, U), f, indent=1, sort_keys=True)
        print("wrote nested_locus.json")
        return
    import telr.TELR_output as O
    if "--only-assembly" in sys.argv:
        with open(os.path.join(GOLD, "prep_assembly.json"), "w") as f:
            json.dump(capture_prep_assembly(), f, indent=1, sort_keys=True)
        print("wrote prep_assembly.json")
        return
    for name, obj in (("prep_assembly.json", capture_prep_assembly()), ("liftover_single.json", capture_liftover(L)), ("liftover_driver.json", capture_liftover_driver(L)),
                      ("af.json", capture_af(T)), ("helpers.json", capture_helpers(L, T, S, U)), ("output.json", capture_output(O)), ("sv.json", capture_sv(S, U)), ("repeatmask.json", capture_repeatmask(T)), ("nested_locus.json", capture_nested(T, L, U))):
        with open(os.path.join(GOLD, name), "w") as f:
            json.dump(obj, f, indent=1, sort_keys=True)
        print("wrote", name)


if __name__ == "__main__":
    main()
