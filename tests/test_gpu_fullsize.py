"""BASELINE configs[1] at FULL size (23.5-Mb genome, 10,000 ONT-like reads, 457 Mbp) on the HIP engine, checked through
size-independent properties, plus bit-exact parity with the oracle on a random sample of the same reads against the same
full-size index:

  * every record's CIGAR consumes exactly its query and target spans, and the match count recomputed from the two
    sequences equals `mlen`; `blen` is the number of alignment columns;
  * the simulated origin of (almost) every read is recovered by its primary record, right strand;
  * idempotence: the same call twice gives the same records and CIGARs;
  * batch independence: the records of a read do not depend on which other reads share its call (the read set mapped as
    two device-side subsets == the read set mapped at once) -- a checksum over checksums of per-read records;
  * a sample of 48 reads mapped by the CPU oracle against the same 23.5-Mb index gives the identical records / CIGARs.
"""
import hashlib

import numpy as np
import pytest

from telr_amd import synth
from telr_amd.presets import preset

pytestmark = pytest.mark.gpu

FIELDS = ["tid", "qlen", "qs", "qe", "tlen", "ts", "te", "mlen", "blen", "score", "subsc", "dp_score", "cnt", "n_sub", "n_cigar", "flags", "mapq"]
_COMP = np.zeros(256, np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b


@pytest.fixture(scope="module")
def full(engine):
    d = synth.make_stage1_dataset(seed=20261002, read_seed=20261002 + 1000)         # the bench's rank-0 data set
    io, mo = preset("map-ont")
    ref_str = bytes(d["ref"]).decode()
    ix = engine.index([ref_str], io)
    qs = engine.seqset(d["reads"])
    res = ix.map(qs, mo)
    return dict(d=d, io=io, mo=mo, ix=ix, qs=qs, res=res, ref_str=ref_str)


def _read(d, i):
    buf, off, ln = d["reads"]
    return buf[off[i]:off[i] + ln[i]]


def _per_read_digest(alns, cigars, qid_of=None):
    """{read id: sha1 over its records (fields + CIGAR ops) in result order}"""
    out = {}
    order = np.argsort(alns["qid"], kind="stable")
    for k in order:
        a = alns[k]
        q = int(a["qid"]) if qid_of is None else int(qid_of[a["qid"]])
        h = out.setdefault(q, hashlib.sha1())
        h.update(np.array([int(a[f]) for f in FIELDS], np.int64).tobytes())
        h.update(np.ascontiguousarray(cigars[a["cigar_off"]:a["cigar_off"] + a["n_cigar"]]).tobytes())
    return {q: h.hexdigest() for q, h in out.items()}


def _digest_of_digests(dg):
    h = hashlib.sha1()
    for q in sorted(dg):
        h.update(("%d:%s;" % (q, dg[q])).encode())
    return h.hexdigest()


def test_cigars_consume_their_spans_and_match_counts(full):
    d, res = full["d"], full["res"]
    ref = d["ref"]
    alns, cig = res.alns, res.cigars
    assert len(alns) >= 10000
    # vectorised span check over all records
    starts = alns["cigar_off"].astype(np.int64); n = alns["n_cigar"].astype(np.int64)
    assert (n > 0).all()
    idx = np.repeat(starts - np.r_[0, np.cumsum(n)[:-1]], n) + np.arange(int(n.sum()))
    ops = cig[idx]; ln = (ops >> 4).astype(np.int64); op = ops & 15
    rec = np.repeat(np.arange(len(alns)), n)
    qcons = np.bincount(rec, weights=ln * ((op == 0) | (op == 1)), minlength=len(alns)).astype(np.int64)
    tcons = np.bincount(rec, weights=ln * ((op == 0) | (op == 2)), minlength=len(alns)).astype(np.int64)
    cols = np.bincount(rec, weights=ln, minlength=len(alns)).astype(np.int64)
    np.testing.assert_array_equal(qcons, alns["qe"] - alns["qs"])
    np.testing.assert_array_equal(tcons, alns["te"] - alns["ts"])
    np.testing.assert_array_equal(cols, alns["blen"])
    assert set(np.unique(op)) <= {0, 1, 2}
    # match counts from the sequences, on every 40th record (python loop over CIGAR ops)
    for k in range(0, len(alns), 40):
        a = alns[k]
        r = _read(d, int(a["qid"]))
        if a["flags"] & 8:
            r = _COMP[r[::-1]]
            q0 = int(a["qlen"] - a["qe"])
        else:
            q0 = int(a["qs"])
        t0, m = int(a["ts"]), 0
        for o in cig[a["cigar_off"]:a["cigar_off"] + a["n_cigar"]]:
            L, c = int(o >> 4), int(o & 15)
            if c == 0:
                m += int((r[q0:q0 + L] == ref[t0:t0 + L]).sum()); q0 += L; t0 += L
            elif c == 1:
                q0 += L
            else:
                t0 += L
        assert m == a["mlen"], (k, m, int(a["mlen"]))


def test_simulated_origins_are_recovered(full):
    d, res = full["d"], full["res"]
    alns = res.alns
    prim = alns[(alns["flags"] & 1) != 0]
    assert len(np.unique(prim["qid"])) == len(prim)                     # one primary per read
    assert len(prim) >= 0.995 * len(d["reads"][2])
    # haplotype -> reference coordinates: subtract what the insertions before the position added
    ins = d["insertions"]
    add = {0: [], 1: []}
    for (p, fam, strand, tsd, af) in ins:
        L = len(d["library"][fam]) + tsd
        add[0].append((p, L))
        if af >= 1.0:
            add[1].append((p, L))

    def to_ref(h, x):
        sh = 0
        for p, L in add[h]:
            if p + sh < x:
                sh += min(L, x - (p + sh))
            else:
                break
        return x - sh
    truth = d["truth"]
    ok = 0
    for a in prim:
        h, s, e, st = (int(v) for v in truth[a["qid"]])
        rs, re = to_ref(h, s), to_ref(h, e)
        if a["ts"] < re + 50 and a["te"] > rs - 50 and ((a["flags"] >> 3) & 1) == st:
            ok += 1
    assert ok >= 0.99 * len(prim), (ok, len(prim))


def _origin_table(d, alns):
    """per primary record: (on the simulated origin?, mapq, fraction of the read's true reference interval that is TE-derived)"""
    prim = alns[(alns["flags"] & 1) != 0]
    ins = d["insertions"]
    add = {0: [], 1: []}
    for (p, fam, strand, tsd, af) in ins:
        L = len(d["library"][fam]) + tsd
        add[0].append((p, L))
        if af >= 1.0:
            add[1].append((p, L))
    cum = {h: (np.array([p for p, _ in add[h]], np.int64), np.array([L for _, L in add[h]], np.int64)) for h in add}

    def to_ref(h, x):
        sh = 0
        for p, L in zip(*cum[h]):
            if p + sh < x:
                sh += min(int(L), x - (int(p) + sh))
            else:
                break
        return x - sh
    te = np.zeros(len(d["ref"]) + 1, np.int32)
    for s_, e_ in d["te_copies"]:
        te[s_] += 1; te[e_] -= 1
    te_cov = np.concatenate([[0], np.cumsum(np.cumsum(te[:-1]) > 0)])          # TE-derived bases before every position
    truth = d["truth"]
    on = np.zeros(len(prim), bool); tef = np.zeros(len(prim))
    for k, a in enumerate(prim):
        h, s, e, st = (int(v) for v in truth[a["qid"]])
        rs, re = to_ref(h, s), to_ref(h, e)
        on[k] = a["ts"] < re + 50 and a["te"] > rs - 50 and ((a["flags"] >> 3) & 1) == st
        tef[k] = (te_cov[min(re, len(te_cov) - 1)] - te_cov[max(rs, 0)]) / max(1, re - rs)
    return prim, on, tef


@pytest.mark.parametrize("name", ["map-ont", "ngmlr-ont", "map-ont/9kb", "ngmlr-pacbio/9kb-clr"])
def test_mapq_calibration_for_the_sniffles_gate(engine, full, name):
    """Sniffles is run with its default minimum mapping quality of 20 (src/telr/TELR_sv.py:49-51; SURVEY hand-off H1): a
    primary record on the read's simulated origin must pass that gate, a primary record somewhere else must not.  Measured
    on all 10,000 reads of configs[1] (15 % of the genome TE-derived), reported for the reads that lie mostly inside
    TE-derived sequence as well."""
    d = full["d"]
    if name == "map-ont":
        alns = full["res"].alns
    elif "/" not in name:
        io, mo = preset(name)
        alns = engine.index([full["ref_str"]], io).map(full["qs"], mo).alns
    else:
        # the read length of configs[2] / [3] (mean 9 kb; many more reads end inside a TE copy) on the same genome
        pname, kind = name.split("/")
        err = (0.013, 0.065, 0.052) if kind.endswith("clr") else (0.04, 0.02, 0.04)
        d = synth.make_stage1_dataset(seed=20261002, n_reads=20000, total_bases=180_000_000, err=err, read_seed=20261002 + 5000)
        assert bytes(d["ref"][:1000]) == bytes(full["d"]["ref"][:1000])
        io, mo = preset(pname)
        alns = engine.index([full["ref_str"]], io).map(engine.seqset(d["reads"]), mo).alns
    prim, on, tef = _origin_table(d, alns)
    q20 = prim["mapq"] >= 20
    right_q20 = (on & q20).sum() / max(1, on.sum())
    wrong_q20 = (~on & q20).sum() / len(prim)
    inte = tef >= 0.5
    rep = {"primaries": int(len(prim)), "on_origin": int(on.sum()), "on_origin_mapq_ge_20": float(right_q20), "wrong_locus_mapq_ge_20_of_all": float(wrong_q20),
           "reads_mostly_TE_derived": int(inte.sum()), "of_those_on_origin": int((on & inte).sum()),
           "of_those_on_origin_mapq_ge_20": float((on & inte & q20).sum() / max(1, (on & inte).sum())),
           "of_those_wrong_locus_mapq_ge_20": int((~on & inte & q20).sum()),
           "mapq_histogram_on_origin": np.bincount(np.minimum(prim["mapq"][on] // 10, 6), minlength=7).tolist(),
           "mapq_histogram_wrong": np.bincount(np.minimum(prim["mapq"][~on] // 10, 6), minlength=7).tolist()}
    print("MAPQ calibration", name, rep)
    assert right_q20 >= 0.97, rep
    assert wrong_q20 <= 0.01, rep


def test_idempotent_and_batch_independent(full):
    ix, qs, mo, res = full["ix"], full["qs"], full["mo"], full["res"]
    whole = _per_read_digest(res.alns, res.cigars)
    again = ix.map(qs, mo)
    assert _digest_of_digests(_per_read_digest(again.alns, again.cigars)) == _digest_of_digests(whole)
    nq = len(full["d"]["reads"][2])
    rng = np.random.default_rng(5)
    perm = rng.permutation(nq)
    parts = {}
    for half in (perm[:nq // 3], perm[nq // 3:]):                       # two unequal, shuffled subsets
        idx = np.sort(half).astype(np.int32)
        sub = qs.subset(idx)
        r = ix.map(sub, mo)
        parts.update(_per_read_digest(r.alns, r.cigars, qid_of=idx))
        sub.free()
    assert set(parts) == set(whole)
    bad = [q for q in whole if parts[q] != whole[q]]
    assert not bad, bad[:10]
    assert _digest_of_digests(parts) == _digest_of_digests(whole)


def test_sample_equals_oracle_on_the_full_index(full):
    from oracle import binding as ob
    d, res, mo, io = full["d"], full["res"], full["mo"], full["io"]
    rng = np.random.default_rng(11)
    ln = d["reads"][2]
    short = np.nonzero(ln < 60000)[0]                                   # keep the CPU side to a few seconds
    pick = np.sort(rng.choice(short, size=48, replace=False))
    oix = ob.OracleIndex([full["ref_str"]], io)
    oref = oix.map([bytes(_read(d, int(i))).decode() for i in pick], mo)
    want = _per_read_digest(oref["alns"], oref["cigars"], qid_of=pick)
    got = _per_read_digest(res.alns, res.cigars)
    for q in pick:
        q = int(q)
        assert got.get(q) == want.get(q), "read %d differs from the oracle" % q


def test_map_pb_hpc_at_full_size(engine, full):
    """the homopolymer-compressed (map-pb) path on the same full-size data: span property over all records and oracle
    parity on a sample of reads against the full-size HPC index"""
    from oracle import binding as ob
    d = full["d"]
    io, mo = preset("map-pb")
    ix = engine.index([full["ref_str"]], io)
    res = ix.map(full["qs"], mo)
    alns, cig = res.alns, res.cigars
    assert len(alns) >= 9900
    n = alns["n_cigar"].astype(np.int64)
    idx = np.repeat(alns["cigar_off"].astype(np.int64) - np.r_[0, np.cumsum(n)[:-1]], n) + np.arange(int(n.sum()))
    ops = cig[idx]; ln = (ops >> 4).astype(np.int64); op = ops & 15
    rec = np.repeat(np.arange(len(alns)), n)
    np.testing.assert_array_equal(np.bincount(rec, weights=ln * (op != 2), minlength=len(alns)).astype(np.int64), alns["qe"] - alns["qs"])
    np.testing.assert_array_equal(np.bincount(rec, weights=ln * (op != 1), minlength=len(alns)).astype(np.int64), alns["te"] - alns["ts"])
    rng = np.random.default_rng(12)
    pick = np.sort(rng.choice(np.nonzero(d["reads"][2] < 50000)[0], size=32, replace=False))
    oref = ob.OracleIndex([full["ref_str"]], io).map([bytes(_read(d, int(i))).decode() for i in pick], mo)
    want = _per_read_digest(oref["alns"], oref["cigars"], qid_of=pick)
    got = _per_read_digest(alns, cig)
    for q in pick:
        assert got.get(int(q)) == want.get(int(q)), "read %d differs from the oracle" % int(q)
    ix.free()
