"""BASELINE configs[2]-[4] shaped inputs on the HIP engine, checked through size-independent properties plus bit-exact
oracle parity on samples (the full runs are `bench.py --config c2|c3|c4`):

  configs[2]  eight dm6-size chromosomes in one index (137.6 Mb + 16-kb pads, close to the 2^31 coordinate space? no:
              2^27.04 -- the TELR_E_RANGE guard is exercised separately), reads from all of them, ranges + long-read lane;
  configs[3]  CLR-like reads (13 % errors 1:5:4) through the NGMLR-style convex-gap preset at configs[1] size;
  configs[4]  a chr22-size target with an 11-Mb leading N block, and the 1,300-family library against every contig with
              per-target ranking (S5 at O(n_loci x library)).
"""
import numpy as np
import pytest

from telr_amd import synth, _lib
from telr_amd._abi import MF_PER_TARGET
from telr_amd.presets import preset
from test_gpu_fullsize import _per_read_digest, _digest_of_digests, _COMP

pytestmark = pytest.mark.gpu


def _spans_ok(alns, cig):
    n = alns["n_cigar"].astype(np.int64)
    assert (n > 0).all()
    idx = np.repeat(alns["cigar_off"].astype(np.int64) - np.r_[0, np.cumsum(n)[:-1]], n) + np.arange(int(n.sum()))
    ops = cig[idx]; ln = (ops >> 4).astype(np.int64); op = ops & 15
    rec = np.repeat(np.arange(len(alns)), n)
    np.testing.assert_array_equal(np.bincount(rec, weights=ln * (op != 2), minlength=len(alns)).astype(np.int64), alns["qe"] - alns["qs"])
    np.testing.assert_array_equal(np.bincount(rec, weights=ln * (op != 1), minlength=len(alns)).astype(np.int64), alns["te"] - alns["ts"])
    np.testing.assert_array_equal(np.bincount(rec, weights=ln, minlength=len(alns)).astype(np.int64), alns["blen"])


def _truth_ok(g, plan, ids, alns, slack=100):
    prim = alns[(alns["flags"] & 1) != 0]
    ok = 0
    for a in prim:
        i = int(ids[a["qid"]])
        c, h = int(plan["chrom"][i]), int(plan["hap"][i])
        s = int(synth.hap_to_ref(g, h, c, plan["start"][i])); e = int(synth.hap_to_ref(g, h, c, plan["start"][i] + plan["length"][i]))
        if a["tid"] == c and a["ts"] < e + slack and a["te"] > s - slack and ((a["flags"] >> 3) & 1) == int(plan["strand"][i]):
            ok += 1
    return ok, len(prim)


def test_c2_shape_many_chromosomes_ranges_and_lane(engine):
    """all eight dm6 arm lengths in one index; 0.6x of reads mapped as ONE call that the engine cuts into several ranges
    (TELR_BATCH_MBP) each with its own long-read lane: same records as the un-split call; origins recovered"""
    import os
    g = synth.make_genome(20261002, synth.DM6_ARMS, n_ins=200, threads=8)
    plan = synth.plan_reads(g, 0.6)
    buf, off, ln, ids = synth.materialize_reads(g, plan, procs=8)
    io, mo = preset("map-ont")
    ix = engine.index([bytes(r).decode() for r in g["ref"]], io)
    n_mz, n_ent = ix.stats()
    assert n_mz > 20_000_000
    qs = engine.seqset((buf, off, ln))
    res = ix.map(qs, mo)
    _spans_ok(res.alns, res.cigars)
    ok, n = _truth_ok(g, plan, ids, res.alns)
    assert n >= 0.99 * len(ln) and ok >= 0.97 * n, (ok, n, len(ln))
    whole = _digest_of_digests(_per_read_digest(res.alns, res.cigars))
    os.environ["TELR_BATCH_MBP"] = "20"; os.environ["TELR_PIPELINE"] = "1"          # several ranges, one at a time
    try:
        res2 = ix.map(qs, mo)
    finally:
        del os.environ["TELR_BATCH_MBP"], os.environ["TELR_PIPELINE"]
    assert _digest_of_digests(_per_read_digest(res2.alns, res2.cigars)) == whole
    assert (np.diff(res2.alns["qid"]) >= 0).all()
    # the same with two ranges in flight (range pipelining; default on calls of 200 Mbp and more)
    os.environ["TELR_BATCH_MBP"] = "12"; os.environ["TELR_PIPELINE"] = "force"
    try:
        res3 = ix.map(qs, mo)
    finally:
        del os.environ["TELR_BATCH_MBP"], os.environ["TELR_PIPELINE"]
    assert _digest_of_digests(_per_read_digest(res3.alns, res3.cigars)) == whole
    assert (np.diff(res3.alns["qid"]) >= 0).all()
    # oracle parity on a few reads against the same 137.6-Mb index would need ~1 min of CPU index build: the full-size
    # parity samples are taken at configs[1] size (test_gpu_fullsize.py); here, a 2-chromosome sub-index suffices
    # to check that target ids / coordinates of a multi-target index agree
    from oracle import binding as ob
    small = [bytes(g["ref"][4]).decode(), bytes(g["ref"][7]).decode()]          # chr4 + chrM
    pick = [i for i in range(len(ln)) if plan["chrom"][ids[i]] in (4, 7)][:40]
    reads = [bytes(buf[off[i]:off[i] + ln[i]]).decode() for i in pick]
    six = engine.index(small, io)
    got = six.map(reads, mo)
    want = ob.OracleIndex(small, io).map(reads, mo)
    assert _digest_of_digests(_per_read_digest(got.alns, got.cigars)) == _digest_of_digests(_per_read_digest(want["alns"], want["cigars"]))


def test_coordinate_range_guard(engine):
    """targets live in one 31-bit coordinate space with 16-kb pads: 131,100 tiny targets exceed it -> TELR_E_RANGE, not a wrap"""
    n = 131100
    seqs = (np.frombuffer(b"ACGTACGTAC" * n, np.uint8), np.arange(n, dtype=np.int64) * 10, np.full(n, 10, np.int32))
    io, _ = preset("map-ont")
    with pytest.raises(_lib.TelrError, match="coordinate range"):
        engine.index(engine.seqset(seqs), io)


def test_c3_shape_clr_reads_convex_gap_preset(engine):
    """configs[3]'s aligner half at configs[1] size: 23.5-Mb genome, CLR-like reads (13 % errors, sub:ins:del 1:5:4), preset
    ngmlr-pacbio ((w,k) = (5,13), convex gap cost as two-piece affine)"""
    from oracle import binding as ob
    g = synth.make_genome(20261002, [("chr2L", 23513712)], n_ins=200, threads=8)
    plan = synth.plan_reads(g, 4.0)
    buf, off, ln, ids = synth.materialize_reads(g, plan, err=(0.013, 0.065, 0.052), procs=8)
    io, mo = preset("ngmlr-pacbio")
    ref = bytes(g["ref"][0]).decode()
    ix = engine.index([ref], io)
    qs = engine.seqset((buf, off, ln))
    res = ix.map(qs, mo)
    _spans_ok(res.alns, res.cigars)
    ok, n = _truth_ok(g, plan, ids, res.alns)
    assert n >= 0.99 * len(ln) and ok >= 0.97 * n, (ok, n, len(ln))
    again = ix.map(qs, mo)
    assert _digest_of_digests(_per_read_digest(again.alns, again.cigars)) == _digest_of_digests(_per_read_digest(res.alns, res.cigars))
    rng = np.random.default_rng(3)
    pick = np.sort(rng.choice(np.nonzero(ln < 40000)[0], size=32, replace=False))
    oref = ob.OracleIndex([ref], io).map([bytes(buf[off[i]:off[i] + ln[i]]).decode() for i in pick], mo)
    want = _per_read_digest(oref["alns"], oref["cigars"], qid_of=pick)
    got = _per_read_digest(res.alns, res.cigars)
    for q in pick:
        assert got.get(int(q)) == want.get(int(q)), "read %d differs from the oracle" % int(q)


def test_c4_shape_n_block_and_big_library_per_target(engine):
    """configs[4]: chr22-size target whose first 11 Mb are N; reads never land in the block; the 1,300-family library against
    the contigs of 150 loci in ONE per-target call finds every locus' family, and equals per-contig calls / the oracle on a sample"""
    from oracle import binding as ob
    g = synth.make_genome(20261002, synth.CHR22, n_fam=1300, n_ins=600, lead_n=11_000_000, threads=8)
    assert (g["ref"][0][:11_000_000] == ord("N")).all() and g["ref"][0][11_000_000] != ord("N")
    plan = synth.plan_reads(g, 1.5)
    buf, off, ln, ids = synth.materialize_reads(g, plan, procs=8)
    io, mo = preset("map-ont")
    ref = bytes(g["ref"][0]).decode()
    ix = engine.index([ref], io)
    qs = engine.seqset((buf, off, ln))
    res = ix.map(qs, mo)
    _spans_ok(res.alns, res.cigars)
    # reads drawn from the N block stay unmapped (or map elsewhere by chance through TE copies); nothing aligns INSIDE it
    assert (res.alns["te"] > 11_000_000 - 1).all() or (res.alns["ts"][res.alns["te"] <= 11_000_000].size == 0)
    assert (res.alns["ts"] >= 11_000_000 - 50).all()
    inside = np.array([plan["start"][ids[i]] + plan["length"][ids[i]] < 10_990_000 for i in range(len(ln))])
    outside = np.array([plan["start"][ids[i]] > 11_010_000 for i in range(len(ln))])
    prim = res.alns[(res.alns["flags"] & 1) != 0]
    mapped = np.zeros(len(ln), bool); mapped[prim["qid"]] = True
    assert mapped[outside].mean() >= 0.99 and mapped[inside].mean() <= 0.01
    ok, n = _truth_ok(g, plan, ids, res.alns)
    assert ok >= 0.97 * n
    # S5: the whole library against every contig, ranked per contig
    loci = synth.make_loci(g, 150)
    contigs = [l["contig"] for l in loci]
    lib = [bytes(x).decode() for x in g["library"]]
    mo5 = mo.copy(); mo5.flags |= MF_PER_TARGET
    cix = engine.index(contigs, io)
    r5 = cix.map(lib, mo5)
    fam_hit = {(int(a["tid"]), int(a["qid"])) for a in r5.alns}
    for k, l in enumerate(loci):
        assert (k, int(l["truth"]["family"][3:])) in fam_hit, l["name"]
    for k in (0, 7, 75, 149):
        solo = engine.index([contigs[k]], io).map(lib, mo)
        osolo = ob.OracleIndex([contigs[k]], io).map(lib, mo)
        sub = r5.alns[r5.alns["tid"] == k]
        assert len(solo.alns) == len(sub) == len(osolo["alns"]) > 0
        for f in ("qid", "qs", "qe", "ts", "te", "mlen", "blen", "dp_score", "score", "cnt", "mapq"):
            np.testing.assert_array_equal(solo.alns[f], osolo["alns"][f], err_msg=f)
            np.testing.assert_array_equal(np.sort(solo.alns[f]), np.sort(sub[f]), err_msg="contig %d field %s" % (k, f))
