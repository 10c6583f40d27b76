"""The call sites that round 1 left unwired, on the HIP engine (through the C ABI):

  S1  alignment(method="nglmr"): the reference's DEFAULT stage-1 aligner (TELR_alignment.py:15-55) = the `ngmlr-*` presets
      ((w,k) = (5,13), NGMLR-like scoring, convex gap as two-piece affine): HIP == oracle, @RG / RG:Z in the sorted BAM;
  S3  the polishing site (TELR_assembly.py:199-236): reads -> draft contig with -r2k, and the primary-only,
      coordinate-sorted, header-less SAM text that `samtools view -F0x900` pipes into wtpoa-cns;
  the `telr-mm2` shim executed for the S3 and S7 argv shapes;
  per-target occurrence cut-offs: one pooled call over hundreds of contigs == one call per contig.
"""
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

from telr_amd import synth, telr_assembly
from telr_amd.fasta import read_fasta
from telr_amd.presets import preset
from test_gpu_parity import compare_all, _read_bgzf, ALN_FIELDS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["ngmlr-ont", "ngmlr-pacbio"])
def test_ngmlr_presets_equal_oracle(engine, data_dir, name):
    _, ts = read_fasta(data_dir + "/ref_38kb.fasta")
    _, qs = read_fasta(data_dir + "/reads.fasta")
    io, mo = preset(name)
    assert (io.k, io.w, io.is_hpc) == (13, 5, 0)
    res, _ = compare_all(engine, ts, qs, io, mo)
    assert len(res.alns) >= 18
    # synthetic reads with the error profile the preset is meant for, repeats and a second target
    rng = np.random.default_rng(77 + len(name))
    genome = [synth.random_seq(rng, 180000), synth.random_seq(rng, 40000)]
    te = synth.random_seq(rng, 2500)
    for g in genome:
        for _ in range(5):
            p = int(rng.integers(0, len(g) - 2500)); g[p:p + 2500] = synth.mutate(rng, te, 0.04, 0.0, 0.0)[:2500]
    err = (0.04, 0.02, 0.04) if name == "ngmlr-ont" else (0.013, 0.065, 0.052)
    reads, truth = synth.simulate_reads(rng, genome, 50, 6000, err=err)
    res, _ = compare_all(engine, genome, reads, io, mo)
    prim = res.alns[(res.alns["flags"] & 1) != 0]
    ok = sum(1 for a in prim if a["tid"] == truth[a["qid"]][0] and a["ts"] < truth[a["qid"]][2] and a["te"] > truth[a["qid"]][1])
    assert ok >= 0.9 * len(truth)


def _bam_records(path):
    raw, _ = _read_bgzf(path)
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].decode()
    off = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, off)[0]; off += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", raw, off)[0]; off += 4
        refs.append(raw[off:off + ln - 1].decode()); off += ln + 4
    recs = []
    while off < len(raw):
        bs = struct.unpack_from("<i", raw, off)[0]
        ref_id, pos, l_name, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", raw, off + 4)
        p = off + 36
        name = raw[p:p + l_name - 1].decode(); p += l_name + 4 * n_cig + (l_seq + 1) // 2 + l_seq
        tags = raw[p:off + 4 + bs]
        recs.append(dict(name=name, ref_id=ref_id, pos=pos, flag=flag, tags=tags))
        off += 4 + bs
    return text, refs, recs


@pytest.mark.parametrize("presets,lb", [("ont", "ont"), ("pacbio", "pb")])
def test_alignment_nglmr_writes_rg_tagged_sorted_bam(engine, data_dir, tmp_path, presets, lb):
    """`alignment(..., method="nglmr", ...)` (sic): ngmlr-* preset, --rg-id/--rg-sm = sample name, --rg-lb ont|pb
    (TELR_alignment.py:32-49), coordinate-sorted BAM + .bai as after sort_index_bam (:103-114)"""
    from telr_amd.telr_alignment import alignment
    bam = str(tmp_path / "s_sort.bam")
    alignment(bam, data_dir + "/reads.fasta", data_dir + "/ref_38kb.fasta", str(tmp_path), "sampleA", 4, "nglmr", presets, engine=engine)
    assert os.path.getsize(bam) > 1000 and os.path.getsize(bam + ".bai") > 30
    text, refs, recs = _bam_records(bam)
    assert "@HD\tVN:1.6\tSO:coordinate" in text
    assert "@RG\tID:sampleA\tSM:sampleA\tLB:%s\n" % lb in text
    assert "CL:ngmlr -r " in text and (" -x %s " % presets) in text
    assert len(recs) >= 18 and all(b"RGZsampleA\x00" in r["tags"] for r in recs)
    keys = [(r["ref_id"] if r["ref_id"] >= 0 else 1 << 30, r["pos"]) for r in recs]
    assert keys == sorted(keys)
    # NGMLR-style output carries MD and no cs
    assert all(b"MDZ" in r["tags"] and b"csZ" not in r["tags"] for r in recs if r["ref_id"] >= 0)
    # the same read set through the oracle with the same preset gives the same number of records per flag class
    from oracle import binding as ob
    _, ts = read_fasta(data_dir + "/ref_38kb.fasta"); _, qs = read_fasta(data_dir + "/reads.fasta")
    io, mo = preset("ngmlr-ont" if presets == "ont" else "ngmlr-pacbio")
    want = ob.OracleIndex(ts, io).map(qs, mo)["alns"]
    for bit, sam in ((4, 0x800), (2, 0x100)):
        assert sum(1 for r in recs if r["flag"] & sam) == int(((want["flags"] & bit) != 0).sum())


def _polish_inputs(n_loci=6, seed=3):
    rng = np.random.default_rng(seed)
    te = synth.random_seq(rng, 3000)
    names, contigs, reads = [], [], []
    for k in range(n_loci):
        c = synth.random_seq(rng, int(rng.integers(15000, 30000)))
        c[6000:9000] = synth.mutate(rng, te, 0.02, 0.0, 0.0)[:3000]          # the same element in every draft contig
        names.append("chr2L_%d_%d" % (1000 * k, 1000 * k + 1)); contigs.append(bytes(c).decode())
        rs = []
        for _ in range(int(rng.integers(10, 25))):
            L = int(rng.integers(3000, 12000)); s0 = int(rng.integers(0, len(c) - L))
            r = c[s0:s0 + L]
            if rng.integers(0, 2):
                r = synth.revcomp_arr(r)
            rs.append(bytes(synth.mutate(rng, r)).decode())
        rs.append(bytes(synth.random_seq(rng, 2000)).decode())                # a read that maps nowhere
        reads.append(rs)
    return names, contigs, reads


def test_polishing_site_s3(engine):
    names, contigs, reads = _polish_inputs()
    sams, alns, cig = telr_assembly.polish_alignments(engine, names, contigs, reads, presets="ont")
    # records: HIP == oracle for the same call (qtarget, bw = 2000)
    io, mo = preset("map-ont"); mo.bw = 2000
    qt = np.array([k for k, rs in enumerate(reads) for _ in rs], np.int32)
    flat = [r for rs in reads for r in rs]
    res, _ = compare_all(engine, contigs, flat, io, mo, qtarget=qt, stages=False)
    for f in ALN_FIELDS:
        np.testing.assert_array_equal(alns[f], res.alns[f], err_msg=f)
    q0 = np.r_[0, np.cumsum([len(rs) for rs in reads])]
    for k, text in enumerate(sams):
        lines = [l.split("\t") for l in text.splitlines()]
        assert lines and not any(l[0].startswith("@") for l in lines)                      # samtools view without -h
        mapped = [l for l in lines if l[2] != "*"]
        assert all(l[2] == names[k] for l in mapped)                                        # only this locus' contig
        assert not any(int(l[1]) & 0x900 for l in lines)                                    # -F0x900
        pos = [int(l[3]) for l in mapped]
        assert pos == sorted(pos)                                                           # samtools sort
        assert [l[2] for l in lines] == [names[k]] * len(mapped) + ["*"] * (len(lines) - len(mapped))   # unmapped last
        # exactly the PRIMARY records of this locus' reads, one per mapped read; SEQ present, CIGAR spans SEQ
        prim = res.alns[((res.alns["flags"] & 1) != 0) & (res.alns["qid"] >= q0[k]) & (res.alns["qid"] < q0[k + 1])]
        assert len(mapped) == len(prim) and sorted(pos) == sorted((prim["ts"] + 1).tolist())
        assert len(lines) == len(reads[k])                                                  # every read appears once
        for l in mapped:
            ops = re.findall(r"(\d+)([MIDSH])", l[5])
            assert sum(int(n) for n, o in ops if o in "MIS") == len(l[9]) and "H" not in l[5]
        assert int(lines[-1][1]) == 4 and lines[-1][0].endswith("_r%d" % (len(reads[k]) - 1))


def test_cli_shim_runs_s3_and_s7(engine, data_dir, tmp_path):
    """`python -m telr_amd.cli_mm2 minimap2 ...` with the reference's own argv for S3 and S7 answers on stdout what the
    in-process calls give (the A/B shim at the reference's subprocess boundary)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    ref, rd = data_dir + "/ref_38kb.fasta", data_dir + "/reads.fasta"
    out = subprocess.run([sys.executable, "-m", "telr_amd.cli_mm2", "minimap2", "-t", "1", "-ax", "map-pb", "-r2k", ref, rd],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if not l.startswith("@")]
    tn, ts = read_fasta(ref); qn, qs = read_fasta(rd)
    io, mo = preset("map-pb"); mo.bw = 2000
    want = engine.index(ts, io).map(qs, mo)
    assert len(lines) == len(want.alns) and "@SQ\tSN:%s" % tn[0] in out.stdout
    assert [int(l.split("\t")[3]) - 1 for l in lines] == want.alns["ts"].tolist()
    # S7: a 500-base flank of the fixture against the fixture, asm10 -N 10 -> PAF
    flank = str(tmp_path / "chr2L_100_200_5p.fa")
    with open(flank, "w") as fh:
        fh.write(">flank\n%s\n" % ts[0][20000:20499])
    out = subprocess.run([sys.executable, "-m", "telr_amd.cli_mm2", "minimap2", "-cx", "asm10", "-v", "0", "-N", "10", ref, flank],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    p = out.stdout.splitlines()[0].split("\t")
    assert p[0] == "flank" and p[4] == "+" and int(p[7]) == 20000 and int(p[8]) == 20499 and int(p[9]) == int(p[10]) == 499
    assert any(t.startswith("cg:Z:499M") for t in p[12:])
    # no hit: rc 0 and empty stdout
    with open(flank, "w") as fh:
        fh.write(">junk\n%s\n" % bytes(synth.random_seq(np.random.default_rng(1), 499)).decode())
    out = subprocess.run([sys.executable, "-m", "telr_amd.cli_mm2", "minimap2", "-cx", "asm10", "-v", "0", "-N", "10", ref, flank],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout == ""


def test_per_target_occurrence_cutoff(engine):
    """S5 over MANY contigs that all carry the same family (TELR_te.py:119-132 runs the aligner once per contig, so a TE
    k-mer is never 'repetitive' there): the pooled PER_TARGET call must give, per contig, what a call against that contig
    alone gives -- and the oracle agrees bit for bit.  With a pooled cut-off the family's minimizers (300 copies) would
    be masked and the annotation lost."""
    rng = np.random.default_rng(2026)
    te = synth.random_seq(rng, 2800); te2 = synth.random_seq(rng, 1500)
    n = 300
    contigs = []
    for k in range(n):
        c = synth.random_seq(rng, int(rng.integers(6000, 9000)))
        x = synth.mutate(rng, te, 0.01, 0.0, 0.0)[:2800]
        c[2000:2000 + len(x)] = x if k % 3 else synth.revcomp_arr(x)
        if k % 50 == 0:                                  # a tandem array inside one contig: ITS own cut-off applies
            for j in range(14):
                c[5200 + 40 * j:5240 + 40 * j] = te2[:40]
        contigs.append(bytes(c).decode())
    lib = [bytes(te).decode(), bytes(te2).decode(), bytes(synth.random_seq(rng, 900)).decode()]
    io, mo = preset("map-ont")
    mo5 = mo.copy(); mo5.flags |= 2
    res, oref = compare_all(engine, contigs, lib, io, mo5, stages=False)
    hit = set(res.alns["tid"][res.alns["qid"] == 0].tolist())
    assert len(hit) == n                                                     # the family is found on every contig
    # one call per contig (what the reference does) for a sample of contigs
    for k in (0, 1, 2, 50, 149, 299):
        solo = engine.index([contigs[k]], io).map(lib, mo)
        sub = res.alns[res.alns["tid"] == k]
        assert len(solo.alns) == len(sub)
        for f in ("qid", "qs", "qe", "ts", "te", "mlen", "blen", "dp_score", "score", "cnt"):
            np.testing.assert_array_equal(np.sort(solo.alns[f]), np.sort(sub[f]), err_msg="contig %d field %s" % (k, f))
    # S4 / S6 shape: a query confined to one target uses that target's cut-off as well
    reads = [contigs[k][1500:5500] for k in range(0, n, 10)]
    qt = np.arange(0, n, 10, dtype=np.int32)
    res2, _ = compare_all(engine, contigs, reads, io, mo, qtarget=qt, stages=False)
    prim = res2.alns[(res2.alns["flags"] & 1) != 0]
    assert len(prim) == len(reads) and ((prim["te"] - prim["ts"]) >= 3900).all()


def test_alignment_called_again_writes_the_same_file(engine, data_dir, tmp_path):
    """alignment() several times in one process, both methods in turn: the prepared output file, the background releases, the
    pooled buffers and the device copy of the CIGARs are reused from call to call -- every call must write the same BAM and
    .bai as the first one of its method (tools/soak_stage1.py does the same on a configs[1]-size read set)"""
    import hashlib
    from telr_amd.telr_alignment import alignment
    seen = {}
    for i in range(6):
        method = "minimap2" if i % 2 == 0 else "nglmr"
        bam = str(tmp_path / ("again%d.bam" % i))
        alignment(bam, data_dir + "/reads.fasta", data_dir + "/ref_38kb.fasta", str(tmp_path), "s", 1, method, "ont", engine=engine)
        h = (hashlib.sha256(open(bam, "rb").read()).hexdigest(), hashlib.sha256(open(bam + ".bai", "rb").read()).hexdigest())
        assert seen.setdefault(method, h) == h, (i, method)
    assert len(seen) == 2 and seen["minimap2"] != seen["nglmr"]


def test_shared_satellite_with_per_query_targets(engine):
    """round 6: a satellite shared by MANY contigs (the hard genome's loci, `bench.py --config c2r`): a satellite k-mer's pooled
    occurrence list runs to tens of thousands of entries, of which a query with its own target (S6, the polishing map) wants its
    target's piece -- found by bisection (`d_occ_lower`), as are the per-target runs of more than eight occurrences in the pooled
    per-target mode (S5).  Every stage against the oracle, per-query targets and MF_PER_TARGET; the reads chain lattices of
    (copies in the read) x (copies in the contig) anchors: runs that take the push loop (kernels.hip.h: WHICH LOOP)."""
    from telr_amd._abi import MF_PER_TARGET
    rng = np.random.default_rng(20261004)
    unit = synth.random_seq(rng, 23)
    contigs = []
    for c in range(12):
        g = synth.random_seq(rng, 9000)
        arr = synth.mutate(rng, np.tile(unit, 130)[:2800], 0.01 * (c % 3), 0.0, 0.0)[:2800]
        p = 2500 + 150 * c
        g[p:p + len(arr)] = arr
        contigs.append(g)
    reads, truth = synth.simulate_reads(rng, contigs, 36, 3500, err=(0.03, 0.015, 0.03))
    qt = np.array([t[0] for t in truth], dtype=np.int32)
    io, mo = preset("map-ont")
    res, oref = compare_all(engine, contigs, reads, io, mo, qtarget=qt)
    assert len(res.alns) >= len(reads) and int(np.diff(oref["anchor_off"]).max()) > 10000      # lattices, not chains
    assert (res.alns["tid"] == qt[res.alns["qid"]]).all()
    mo5 = mo.copy(); mo5.flags |= MF_PER_TARGET
    res5, _ = compare_all(engine, contigs, reads, io, mo5)
    assert len(res5.alns) >= len(res.alns)
