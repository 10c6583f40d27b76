#!/usr/bin/env python3
"""Cross-check against the UPSTREAM tools of the reference pipeline, when a box has them on PATH.

SURVEY.md section 7 (hard part 1) / 8(d) and BASELINE.md section 2: minimap2 2.22, ngmlr 0.2.7, samtools and bedtools
(envs/telr.yml:45-48) are absent from /root/reference, from the build image and from the GPU box, so the aligner arithmetic
of this repository is "parity unpinned".  This module is the route out of that on any machine that does have them:

  * `find_tools()`            shutil.which() for minimap2 / ngmlr / samtools / bedtools;
  * S1 / S2 / S7              the reference's own argv shapes, run exactly as TELR_alignment.py:31-51 (ngmlr), :69-82
                              (minimap2 stage 1) and TELR_liftover.py:253-266 (flanks, asm10 -N 10) run them -- same flags,
                              same order, stdout redirected into a file -- on the bundled fixture (tests/data) and, from
                              bench.py, on a bounded sample of the bench's own read set; wall clock and threads stated
                              (= the "reference CPU path" timed on the box's host cores);
  * drift table               every record of the upstream SAM / PAF against the records `telr_amd.cli_mm2` writes for the
                              SAME argv and files: reads placed at all, primary placements (target, strand) agreeing, start
                              / end coordinates identical, within 10 bases, CIGAR identical, MAPQ identical;
  * bedtools                  `closest -s -d -t all`, `closest -d -D ref -k K`, `merge -d D -c cols -o distinct -delim X`,
                              `intersect -wao`, `sort` of the real binary against `telr_amd/intervals.py` AND against the
                              hand-derived expectations of tests/golden/bedtools_handmade.json;
  * samtools                  `samtools quickcheck` + `samtools view -c` on a BAM this repository wrote.

When nothing is on PATH the answer is `{"available": false, "looked_for": [...]}` -- printed, never silently skipped.

What a run on a box WITH the tools settles (DESIGN section 2 lists the same): the long join at S5 (does minimap2 2.22 chain a
library hit across an insertion nested in a reference TE copy: 914 vs 968 recovered loci), the MAPQ formula, the tie order
of `closest -t all` / `merge -o distinct`, NGMLR's segmentation against the minimap2-style chain DP of the `ngmlr-*` presets.

usage:  python tools/crosscheck_upstream.py [--out DIR] [--threads N] [--no-engine]      (prints ONE JSON object)
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LOOKED_FOR = ("minimap2", "ngmlr", "samtools", "bedtools")


def find_tools():
    return {t: shutil.which(t) for t in LOOKED_FOR}


# ---- the reference's argv shapes, verbatim -------------------------------------------------------------------------------
def argv_s1(reference, read, presets, thread, sample_name):
    """TELR_alignment.py:31-51 (`method == "nglmr"`)"""
    label = {"ont": "ont", "pacbio": "pb"}[presets]
    return ["ngmlr", "-r", reference, "-q", read, "-x", presets, "-t", str(thread), "--rg-id", sample_name, "--rg-sm", sample_name,
            "--rg-lb", label, "--no-progress"]


def argv_s2(reference, read, presets):
    """TELR_alignment.py:69-82 (`method == "minimap2"`): no -t, i.e. minimap2's default of 3 worker threads"""
    return ["minimap2", "--cs", "--MD", "-Y", "-L", "-ax", {"ont": "map-ont", "pacbio": "map-pb"}[presets], reference, read]


def argv_s7(ref_fa, flank_fa, preset="asm10", num_secondary=10):
    """TELR_liftover.py:253-266"""
    return ["minimap2", "-cx", preset, "-v", "0", "-N", str(num_secondary), ref_fa, flank_fa]


def threads_of(argv):
    if "-t" in argv:
        return int(argv[argv.index("-t") + 1])
    return 3 if argv[0].endswith("minimap2") else 1          # minimap2's default -t 3


def run_upstream(argv, out_path, timeout=3600):
    """the reference's `with open(out) as output: subprocess.call(argv, stdout=output)`, timed"""
    t0 = time.time()
    with open(out_path, "w") as output:
        rc = subprocess.call(argv, stdout=output, stderr=subprocess.DEVNULL, timeout=timeout)
    return {"argv": argv, "seconds": time.time() - t0, "threads": threads_of(argv), "exit_code": rc}


# ---- SAM / PAF -> comparable records -------------------------------------------------------------------------------------
_CIG = re.compile(r"(\d+)([MIDNSHP=X])")


def _cigar_spans(cigar):
    """-> (reference bases, query bases aligned, leading clip, trailing clip, CIGAR with =/X folded into M and clips dropped)"""
    ops = [(int(n), o) for n, o in _CIG.findall(cigar)]
    lead = trail = 0
    if ops and ops[0][1] in "SH":
        lead = ops[0][0]
    if len(ops) > 1 and ops[-1][1] in "SH":
        trail = ops[-1][0]
    core = []
    for n, o in ops:
        if o in "SHP":
            continue
        o = "M" if o in "=X" else o
        if core and core[-1][1] == o:
            core[-1] = (core[-1][0] + n, o)
        else:
            core.append((n, o))
    rlen = sum(n for n, o in core if o in "MDN")
    qaln = sum(n for n, o in core if o in "MI")
    return rlen, qaln, lead, trail, "".join("%d%s" % x for x in core)


def parse_sam(path):
    """-> list of dict(q, t, strand, ts, te, qs, qe [on the read's forward strand], mapq, cigar, primary, supplementary); unmapped lines dropped"""
    out = []
    with open(path) as fh:
        for line in fh:
            if line.startswith("@"):
                continue
            f = line.rstrip("\n").split("\t")
            if len(f) < 11:
                continue
            flag = int(f[1])
            if flag & 4 or f[2] == "*":
                continue
            rlen, qaln, lead, trail, core = _cigar_spans(f[5])
            rev = bool(flag & 16)
            qlen = lead + qaln + trail
            qs = trail if rev else lead
            out.append(dict(q=f[0], t=f[2], strand="-" if rev else "+", ts=int(f[3]) - 1, te=int(f[3]) - 1 + rlen, qs=qs, qe=qs + qaln, qlen=qlen,
                            mapq=int(f[4]), cigar=core, primary=not (flag & 0x900), supplementary=bool(flag & 0x800)))
    return out


def parse_paf(path):
    out = []
    with open(path) as fh:
        for line in fh:
            f = line.rstrip("\n").split("\t")
            if len(f) < 12:
                continue
            tags = {x[:2]: x[5:] for x in f[12:]}
            cg = tags.get("cg", "")
            core = _cigar_spans(cg)[4] if cg else ""
            out.append(dict(q=f[0], t=f[5], strand=f[4], ts=int(f[7]), te=int(f[8]), qs=int(f[2]), qe=int(f[3]), qlen=int(f[1]), mapq=int(f[11]),
                            cigar=core, primary=tags.get("tp", "P") == "P", supplementary=False))
    return out


def drift(up, ours, near=10):
    """record-level drift of two record lists (upstream, ours) -> dict of counts and fractions.
    Records are matched per read: first by (target, strand) and best reciprocal overlap on the target."""
    def by(rs):
        d = {}
        for r in rs:
            d.setdefault(r["q"], []).append(r)
        return d
    U, O = by(up), by(ours)
    reads = sorted(set(U) | set(O))
    n = dict(reads=len(reads), reads_only_upstream=0, reads_only_ours=0, records_upstream=len(up), records_ours=len(ours), matched=0, unmatched_upstream=0,
             unmatched_ours=0, coords_identical=0, coords_within=0, cigar_identical=0, mapq_identical=0, primary_same_place=0, primary_compared=0)
    for q in reads:
        u, o = U.get(q, []), O.get(q, [])
        if not o:
            n["reads_only_upstream"] += 1
        if not u:
            n["reads_only_ours"] += 1
        used = set()
        for a in u:
            best, bj = 0, -1
            for j, b in enumerate(o):
                if j in used or b["t"] != a["t"] or b["strand"] != a["strand"]:
                    continue
                ov = min(a["te"], b["te"]) - max(a["ts"], b["ts"])
                if ov > best:
                    best, bj = ov, j
            if bj < 0:
                n["unmatched_upstream"] += 1
                continue
            used.add(bj)
            b = o[bj]
            n["matched"] += 1
            same = (a["ts"], a["te"], a["qs"], a["qe"]) == (b["ts"], b["te"], b["qs"], b["qe"])
            n["coords_identical"] += same
            n["coords_within"] += max(abs(a["ts"] - b["ts"]), abs(a["te"] - b["te"])) <= near
            n["cigar_identical"] += bool(same and a["cigar"] and a["cigar"] == b["cigar"])
            n["mapq_identical"] += a["mapq"] == b["mapq"]
        n["unmatched_ours"] += len(o) - len(used)
        pu = [r for r in u if r["primary"] and not r["supplementary"]]
        po = [r for r in o if r["primary"] and not r["supplementary"]]
        if pu and po:
            n["primary_compared"] += 1
            a, b = pu[0], po[0]
            n["primary_same_place"] += a["t"] == b["t"] and a["strand"] == b["strand"] and min(a["te"], b["te"]) > max(a["ts"], b["ts"])
    m = max(1, n["matched"])
    n["frac_records_unmatched"] = (n["unmatched_upstream"] + n["unmatched_ours"]) / max(1, n["records_upstream"] + n["records_ours"])
    n["frac_coords_identical"] = n["coords_identical"] / m
    n["frac_coords_within_%d" % near] = n["coords_within"] / m
    n["frac_cigar_identical"] = n["cigar_identical"] / m
    n["frac_mapq_identical"] = n["mapq_identical"] / m
    n["frac_primary_same_place"] = n["primary_same_place"] / max(1, n["primary_compared"])
    return n


# ---- one argv shape: upstream vs ours ------------------------------------------------------------------------------------
def default_ours(engine=None):
    """the engine behind the same argv: telr_amd.cli_mm2.run (fails loudly without the HIP library / a device)"""
    from telr_amd import cli_mm2

    def ours(argv, out_path):
        t0 = time.time()
        cli_mm2.run(argv, out_path, engine=engine)
        return time.time() - t0
    return ours


def crosscheck_shape(name, argv, workdir, ours=None, what=""):
    """run `argv` with the upstream tool on PATH and with `ours(argv, out)`; -> dict(upstream timing, ours timing, drift)"""
    is_sam = argv[0].endswith("ngmlr") or "-ax" in argv or "-a" in argv
    ext = ".sam" if is_sam else ".paf"
    up_out = os.path.join(workdir, name + ".upstream" + ext)
    res = {"shape": name, "what": what, "upstream": run_upstream(argv, up_out)}
    parse = parse_sam if is_sam else parse_paf
    up = parse(up_out)
    res["upstream"]["records"] = len(up)
    if ours is not None:
        our_out = os.path.join(workdir, name + ".ours" + ext)
        try:
            dt = ours(argv, our_out)
            mine = parse(our_out)
            res["ours"] = {"seconds_incl_index_build_and_file_io": dt, "records": len(mine)}
            res["drift"] = drift(up, mine)
        except Exception as e:
            res["ours"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


# ---- bedtools ------------------------------------------------------------------------------------------------------------
def _write_bed(path, rows):
    with open(path, "w") as fh:
        for r in rows:
            fh.write("\t".join(map(str, r)) + "\n")


def bedtools_argv(case, a_path, b_path):
    """the reference's bedtools invocations for the five families of hand-made cases (TELR_liftover.py:244,306-324,501-518;
    TELR_te.py:149-160,196-206,330)"""
    t, args = case["tool"], case.get("args", {})
    if t == "closest_s_d_tall":
        return ["bedtools", "closest", "-a", a_path, "-b", b_path, "-s", "-d", "-t", "all"]
    if t == "closest_D_ref_k":
        return ["bedtools", "closest", "-a", a_path, "-b", b_path, "-d", "-D", "ref", "-k", str(args["k"])]
    if t == "merge_distinct":
        cols = args["cols"]
        return ["bedtools", "merge", "-d", str(args["d"]), "-c", ",".join(str(c + 1) for c in cols), "-o", ",".join(["distinct"] * len(cols)), "-delim", args["delim"], "-i", a_path]
    if t == "intersect_wao":
        return ["bedtools", "intersect", "-a", a_path, "-b", b_path, "-wao"]
    return ["bedtools", "sort", "-i", a_path]


def intervals_answer(case):
    from telr_amd import intervals as iv
    a, b, args, t = case["a"], case.get("b"), case.get("args", {}), case["tool"]
    if t == "closest_s_d_tall":
        got = iv.closest_same_strand(a, b)
    elif t == "closest_D_ref_k":
        got = iv.closest_signed_k(a, b, k=args["k"])
    elif t == "merge_distinct":
        got = iv.merge_distinct(a, args["d"], args["cols"], args["delim"])
    elif t == "intersect_wao":
        got = iv.intersect_wao(a, b)
    else:
        got = iv.bed_sort(a)
    return [list(map(str, r)) for r in got]


def crosscheck_bedtools(workdir, golden=None):
    golden = golden or os.path.join(ROOT, "tests", "golden", "bedtools_handmade.json")
    cases = json.load(open(golden))["cases"]
    rows = []
    for k, c in enumerate(cases):
        a_path, b_path = os.path.join(workdir, "bt%d.a.bed" % k), os.path.join(workdir, "bt%d.b.bed" % k)
        _write_bed(a_path, c["a"])
        _write_bed(b_path, c.get("b") or [])
        argv = bedtools_argv(c, a_path, b_path)
        p = subprocess.run(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        real = [ln.split("\t") for ln in p.stdout.splitlines() if ln]
        mine = intervals_answer(c)
        rows.append({"case": c["tool"] + ":" + c["name"], "argv": argv[:2] + [x for x in argv[2:] if not x.startswith(workdir)], "exit_code": p.returncode,
                     "bedtools_equals_intervals_py": real == mine, "bedtools_equals_hand_derived": real == c["expected"],
                     "bedtools_rows": len(real), "intervals_py_rows": len(mine)})
    return {"cases": len(rows), "bedtools_equals_intervals_py": sum(r["bedtools_equals_intervals_py"] for r in rows),
            "bedtools_equals_hand_derived": sum(r["bedtools_equals_hand_derived"] for r in rows),
            "differing": [r for r in rows if not (r["bedtools_equals_intervals_py"] and r["bedtools_equals_hand_derived"])]}


# ---- samtools on a BAM of this repository ----------------------------------------------------------------------------------
def crosscheck_samtools(bam_path, n_records=None):
    out = {"bam": os.path.basename(bam_path)}
    out["quickcheck_exit_code"] = subprocess.call(["samtools", "quickcheck", bam_path])
    p = subprocess.run(["samtools", "view", "-c", bam_path], stdout=subprocess.PIPE, text=True)
    out["view_c"] = int(p.stdout.strip() or -1) if p.returncode == 0 else None
    if n_records is not None:
        out["records_written"] = n_records
        out["count_matches"] = out["view_c"] == n_records
    if os.path.exists(bam_path + ".bai"):
        p = subprocess.run(["samtools", "idxstats", bam_path], stdout=subprocess.PIPE, text=True)
        out["idxstats_exit_code"] = p.returncode
        out["idxstats_mapped"] = sum(int(l.split("\t")[2]) for l in p.stdout.splitlines() if l.count("\t") >= 3)
    return out


# ---- the whole leg ---------------------------------------------------------------------------------------------------------
def _write_fasta(path, names, seqs):
    with open(path, "w") as fh:
        for n, s in zip(names, seqs):
            fh.write(">%s\n%s\n" % (n, s if isinstance(s, str) else bytes(s).decode()))


def reference_cpu_path(ours=None, sample=None, threads=None, workdir=None, presets="ont", bam_writer=None):
    """The `reference_cpu_path` object of the bench line.  `sample` = dict(ref_names, ref_seqs, read_names, read_seqs, flank_names,
    flank_seqs, text) or a callable returning it (built only when a tool was found): a bounded sample of the bench's own workload
    (None: the bundled fixture only).  `bam_writer(ref_fa, reads_fa, bam_path) -> records`: this repository's stage-1 hand-off on
    the fixture, for `samtools quickcheck / view -c / idxstats`."""
    tools = find_tools()
    out = {"available": any(tools.values()), "looked_for": list(LOOKED_FOR), "found": {k: v for k, v in tools.items() if v}}
    if not out["available"]:
        out["note"] = ("none of the reference's external tools is on PATH on this box: the reference CPU path (ngmlr + minimap2, TELR_alignment.py:31-82, "
                       "TELR_liftover.py:253-266) cannot be timed or compared here; cpu_baseline (the C oracle, kind \"port\") is the CPU figure of this line")
        return out
    threads = threads or max(1, len(os.sched_getaffinity(0)))
    own = workdir is None
    workdir = workdir or tempfile.mkdtemp(prefix="telr_xcheck_")
    try:
        data = os.path.join(ROOT, "tests", "data")
        sets = [("fixture", os.path.join(data, "ref_38kb.fasta"), os.path.join(data, "reads.fasta"), None,
                 "the reference's bundled test/ref_38kb.fasta + test/reads.fasta (BASELINE configs[0])")]
        if callable(sample):
            sample = sample()
        if sample is not None:
            rf, qf, ff = (os.path.join(workdir, n) for n in ("sample_ref.fa", "sample_reads.fa", "sample_flanks.fa"))
            _write_fasta(rf, sample["ref_names"], sample["ref_seqs"])
            _write_fasta(qf, sample["read_names"], sample["read_seqs"])
            if sample.get("flank_seqs"):
                _write_fasta(ff, sample["flank_names"], sample["flank_seqs"])
            else:
                ff = None
            sets.append(("sample", rf, qf, ff, sample.get("text", "a sample of the bench's read set")))
        out["cores"] = threads
        out["shapes"] = []
        for tag, rf, qf, ff, text in sets:
            if tools["ngmlr"]:
                out["shapes"].append(crosscheck_shape("S1_%s" % tag, argv_s1(rf, qf, presets, threads, "xcheck"), workdir, ours, "stage 1, the reference's default aligner; " + text))
            if tools["minimap2"]:
                out["shapes"].append(crosscheck_shape("S2_%s" % tag, argv_s2(rf, qf, presets), workdir, ours, "stage 1, --aligner minimap2; " + text))
                if ff:
                    out["shapes"].append(crosscheck_shape("S7_%s" % tag, argv_s7(rf, ff), workdir, ours, "flank -> reference, asm10 -N 10; " + text))
        for s in out["shapes"]:
            if s["shape"].endswith("_sample") and s["upstream"]["exit_code"] == 0 and sample is not None and s["shape"][:2] in ("S1", "S2"):
                s["upstream"]["gbp_per_s"] = sample.get("read_bases", 0) / max(1e-9, s["upstream"]["seconds"]) / 1e9
        if tools["bedtools"]:
            out["bedtools"] = crosscheck_bedtools(workdir)
        if tools["samtools"] and bam_writer is not None:
            bam_path = os.path.join(workdir, "fixture.bam")
            try:
                out["samtools"] = crosscheck_samtools(bam_path, bam_writer(sets[0][1], sets[0][2], bam_path))
            except Exception as e:
                out["samtools"] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["settles"] = ["long join at S5 (914 vs 968 recovered loci, DESIGN 3.11)", "MAPQ formula (DESIGN 3.8)", "tie order of `closest -t all` / `merge -o distinct`",
                          "NGMLR segmentation vs the chain DP of the ngmlr-* presets (DESIGN 3.9)"]
    finally:
        if own:
            shutil.rmtree(workdir, ignore_errors=True)
    return out


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="", help="keep the SAM / PAF files of both sides in this directory")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--presets", default="ont", choices=["ont", "pacbio"])
    ap.add_argument("--no-engine", action="store_true", help="time the upstream tools only (no device on this box)")
    a = ap.parse_args()
    if a.out:
        os.makedirs(a.out, exist_ok=True)
    ours = None
    if not a.no_engine and any(find_tools().values()):
        ours = default_ours()
    print(json.dumps(reference_cpu_path(ours=ours, threads=a.threads or None, workdir=a.out or None, presets=a.presets)))


if __name__ == "__main__":
    main()
