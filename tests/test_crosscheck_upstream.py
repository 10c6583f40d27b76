"""tools/crosscheck_upstream.py: the upstream cross-check leg, exercised on CPU with FAKE
`minimap2` / `ngmlr` / `bedtools` / `samtools` executables on PATH (shell / python scripts that print canned SAM / PAF): detection,
the reference's argv shapes, SAM / PAF parsing and the drift table.  The real comparison needs a box that has the tools."""
import json
import os
import stat
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import crosscheck_upstream as xc          # noqa: E402

SAM_UP = "\n".join([
    "@HD\tVN:1.6\tSO:unsorted",
    "@SQ\tSN:ref\tLN:38000",
    "r1\t0\tref\t101\t60\t5S40M2I30M3D20M10S\t*\t0\t0\t*\t*",
    "r2\t16\tref\t2001\t60\t100M\t*\t0\t0\t*\t*",
    "r2\t2064\tref\t9001\t20\t60H40M\t*\t0\t0\t*\t*",
    "r3\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*",
    "r4\t0\tref\t501\t37\t50M\t*\t0\t0\t*\t*",
]) + "\n"
# ours: r1 identical (written with =/X), r2 primary ends 4 bases later, its supplementary is missing, r4 identical with another MAPQ, r5 only here
SAM_OURS = "\n".join([
    "@SQ\tSN:ref\tLN:38000",
    "r1\t0\tref\t101\t60\t5S30=1X9=2I30=3D20=10S\t*\t0\t0\t*\t*",
    "r2\t16\tref\t2001\t60\t104M\t*\t0\t0\t*\t*",
    "r4\t0\tref\t501\t12\t50M\t*\t0\t0\t*\t*",
    "r5\t0\tref\t7001\t60\t80M\t*\t0\t0\t*\t*",
]) + "\n"
PAF_UP = "f1\t500\t0\t500\t+\tref\t38000\t1000\t1500\t498\t500\t60\ttp:A:P\tcg:Z:500M\n" \
         "f2\t500\t10\t500\t-\tref\t38000\t3000\t3490\t480\t490\t0\ttp:A:S\tcg:Z:490M\n"


def _fake(dirname, name, body):
    p = os.path.join(dirname, name)
    with open(p, "w") as fh:
        fh.write("#!%s\n" % sys.executable + body)
    os.chmod(p, os.stat(p).st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
    return p


def test_parse_and_drift():
    import tempfile
    d = tempfile.mkdtemp()
    up, ours = os.path.join(d, "u.sam"), os.path.join(d, "o.sam")
    open(up, "w").write(SAM_UP); open(ours, "w").write(SAM_OURS)
    u, o = xc.parse_sam(up), xc.parse_sam(ours)
    assert len(u) == 4 and len(o) == 4                                  # the unmapped line is dropped
    r1 = u[0]
    assert (r1["ts"], r1["te"], r1["qs"], r1["qe"], r1["qlen"]) == (100, 100 + 40 + 30 + 3 + 20, 5, 5 + 92, 107) and r1["cigar"] == "40M2I30M3D20M"
    assert o[0]["cigar"] == r1["cigar"]                                 # =/X folded into M
    r2 = u[1]
    assert r2["strand"] == "-" and (r2["qs"], r2["qe"]) == (0, 100)
    assert u[2]["supplementary"] and not u[2]["primary"] and (u[2]["qs"], u[2]["qe"]) == (0, 40)        # reverse strand: the clip is at the read's end
    t = xc.drift(u, o)
    assert t["reads"] == 4 and t["reads_only_ours"] == 1 and t["reads_only_upstream"] == 0
    assert t["matched"] == 3 and t["unmatched_upstream"] == 1 and t["unmatched_ours"] == 1
    assert t["coords_identical"] == 2 and t["coords_within"] == 3 and t["cigar_identical"] == 2 and t["mapq_identical"] == 2
    assert t["primary_compared"] == 3 and t["primary_same_place"] == 3
    paf = os.path.join(d, "u.paf"); open(paf, "w").write(PAF_UP)
    p = xc.parse_paf(paf)
    assert len(p) == 2 and p[0]["primary"] and not p[1]["primary"] and p[1]["strand"] == "-" and p[0]["cigar"] == "500M"


def test_argv_shapes_are_the_references():
    assert xc.argv_s1("ref.fa", "reads.fa", "ont", 8, "S") == ["ngmlr", "-r", "ref.fa", "-q", "reads.fa", "-x", "ont", "-t", "8", "--rg-id", "S", "--rg-sm", "S",
                                                              "--rg-lb", "ont", "--no-progress"]
    assert xc.argv_s1("r", "q", "pacbio", 1, "S")[-2] == "pb"
    assert xc.argv_s2("ref.fa", "reads.fa", "pacbio") == ["minimap2", "--cs", "--MD", "-Y", "-L", "-ax", "map-pb", "ref.fa", "reads.fa"]
    assert xc.argv_s7("ref.fa", "flank.fa") == ["minimap2", "-cx", "asm10", "-v", "0", "-N", "10", "ref.fa", "flank.fa"]
    # every shape is one cli_mm2 accepts
    from telr_amd.cli_mm2 import parse_argv
    for av in (xc.argv_s1("r", "q", "ont", 4, "S"), xc.argv_s2("r", "q", "ont"), xc.argv_s7("r", "q")):
        assert parse_argv(av)["target"] == "r"
    assert xc.threads_of(xc.argv_s2("r", "q", "ont")) == 3 and xc.threads_of(xc.argv_s1("r", "q", "ont", 6, "S")) == 6


def test_absent_tools_say_so(monkeypatch, tmp_path):
    monkeypatch.setenv("PATH", str(tmp_path))
    out = xc.reference_cpu_path()
    assert out["available"] is False and out["looked_for"] == ["minimap2", "ngmlr", "samtools", "bedtools"] and "note" in out
    json.dumps(out)


def test_leg_with_fake_tools_on_path(monkeypatch, tmp_path):
    bindir = tmp_path / "bin"; bindir.mkdir()
    # fake aligners: SAM for -ax / ngmlr, PAF for -cx; they log their argv
    log = tmp_path / "argv.log"
    body = "import sys\nopen(%r,'a').write(' '.join(sys.argv)+'\\n')\nsys.stdout.write(%%r)\n" % str(log)
    _fake(str(bindir), "ngmlr", body % SAM_UP)
    _fake(str(bindir), "minimap2", "import sys\nopen(%r,'a').write(' '.join(sys.argv)+'\\n')\nsys.stdout.write(%r if '-ax' in sys.argv else %r)\n" % (str(log), SAM_UP, PAF_UP))
    # fake bedtools: answers with telr_amd/intervals.py (plumbing only -- this is NOT evidence about bedtools)
    _fake(str(bindir), "bedtools", "import sys, json\nsys.path.insert(0, %r); sys.path.insert(0, %r)\nimport crosscheck_upstream as xc\n"
          "a = sys.argv\nrd = lambda p: [l.rstrip('\\n').split('\\t') for l in open(p) if l.strip()]\n"
          "sub = a[1]\n"
          "if sub == 'closest':\n"
          "    A, B = rd(a[a.index('-a') + 1]), rd(a[a.index('-b') + 1])\n"
          "    c = {'tool': 'closest_D_ref_k' if '-k' in a else 'closest_s_d_tall', 'a': A, 'b': B, 'args': {'k': int(a[a.index('-k') + 1])} if '-k' in a else {}}\n"
          "elif sub == 'merge':\n"
          "    c = {'tool': 'merge_distinct', 'a': rd(a[a.index('-i') + 1]), 'args': {'d': int(a[a.index('-d') + 1]), 'cols': [int(x) - 1 for x in a[a.index('-c') + 1].split(',')], 'delim': a[a.index('-delim') + 1]}}\n"
          "elif sub == 'intersect':\n"
          "    c = {'tool': 'intersect_wao', 'a': rd(a[a.index('-a') + 1]), 'b': rd(a[a.index('-b') + 1]), 'args': {}}\n"
          "else:\n"
          "    c = {'tool': 'sort', 'a': rd(a[a.index('-i') + 1]), 'args': {}}\n"
          "for r in xc.intervals_answer(c):\n    print('\\t'.join(r))\n" % (ROOT, os.path.join(ROOT, "tools")))
    monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ.get("PATH", ""))

    def ours(argv, out_path):                      # stands for telr_amd.cli_mm2.run on a box with a device
        is_sam = argv[0] == "ngmlr" or "-ax" in argv
        open(out_path, "w").write(SAM_OURS if is_sam else PAF_UP)
        return 0.01
    sample = dict(ref_names=["ref"], ref_seqs=["ACGT" * 200], read_names=["r1", "r2"], read_seqs=["ACGT" * 50, "TTGA" * 40], flank_names=["f1"], flank_seqs=["ACGT" * 100],
                  read_bases=360, text="unit-test sample")
    work = tmp_path / "work"; work.mkdir()
    out = xc.reference_cpu_path(ours=ours, sample=sample, threads=4, workdir=str(work))
    json.dumps(out)
    assert out["available"] and set(out["found"]) == {"minimap2", "ngmlr", "bedtools"} and out["cores"] == 4
    shapes = {s["shape"]: s for s in out["shapes"]}
    assert set(shapes) == {"S1_fixture", "S2_fixture", "S1_sample", "S2_sample", "S7_sample"}
    s1 = shapes["S1_sample"]
    assert s1["upstream"]["exit_code"] == 0 and s1["upstream"]["threads"] == 4 and s1["upstream"]["records"] == 4 and s1["ours"]["records"] == 4
    assert s1["drift"]["matched"] == 3 and s1["drift"]["coords_identical"] == 2 and "gbp_per_s" in s1["upstream"]
    assert shapes["S7_sample"]["drift"]["frac_coords_identical"] == 1.0 and shapes["S7_sample"]["drift"]["frac_records_unmatched"] == 0.0
    # the binaries saw the reference's argv, flag for flag
    seen = open(str(log)).read().splitlines()
    assert any(l.split()[1:6] == ["--cs", "--MD", "-Y", "-L", "-ax"] for l in seen)
    assert any(" -x ont -t 4 --rg-id xcheck --rg-sm xcheck --rg-lb ont --no-progress" in l for l in seen)
    assert any(l.split()[1:7] == ["-cx", "asm10", "-v", "0", "-N", "10"] for l in seen)
    bt = out["bedtools"]
    n_cases = len(json.load(open(os.path.join(ROOT, "tests", "golden", "bedtools_handmade.json")))["cases"])          # (39 since round 6)
    assert bt["cases"] == n_cases and bt["bedtools_equals_intervals_py"] == n_cases and bt["bedtools_equals_hand_derived"] == n_cases and not bt["differing"]


def test_samtools_leg_with_a_fake_binary(monkeypatch, tmp_path):
    """`samtools quickcheck / view -c / idxstats` on a BAM the library wrote (here: a stand-in file and a fake samtools that answers like one)"""
    bindir = tmp_path / "bin"; bindir.mkdir()
    _fake(str(bindir), "samtools", "import sys\na = sys.argv[1:]\n"
          "if a[0] == 'quickcheck': sys.exit(0)\n"
          "if a[0] == 'view' and a[1] == '-c': print(19); sys.exit(0)\n"
          "if a[0] == 'idxstats': print('ref\\t38000\\t19\\t0'); print('*\\t0\\t0\\t6'); sys.exit(0)\n"
          "sys.exit(2)\n")
    monkeypatch.setenv("PATH", str(bindir))
    work = tmp_path / "work"; work.mkdir()

    def bam_writer(ref_fa, reads_fa, bam_path):
        assert ref_fa.endswith("ref_38kb.fasta") and reads_fa.endswith("reads.fasta")
        open(bam_path, "wb").write(b"BAM\1"); open(bam_path + ".bai", "wb").write(b"BAI\1")
        return 19
    out = xc.reference_cpu_path(ours=None, threads=2, workdir=str(work), bam_writer=bam_writer)
    json.dumps(out)
    assert out["available"] and list(out["found"]) == ["samtools"] and out["shapes"] == []
    st = out["samtools"]
    assert st["quickcheck_exit_code"] == 0 and st["view_c"] == 19 and st["count_matches"] is True and st["idxstats_mapped"] == 19
