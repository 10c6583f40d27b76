"""Gate on the speed-tuned engine spec: the presets' bounded look-back (128 anchors), adaptive
first-pass band (2 + q4*sqrt(mn)/16, retried with a wider band when the path touches the edge) and capped end
extension (ext_max = 2048 bases in a band of +-31) are compared, record by record, with the oracle's FAITHFUL mode --
gap fills over the whole -r band, extensions over the whole rest of the read in a band of -r diagonals, look-back 5000
(minimap2's max_chain_iter; Li 2018 section 2.1-2.3) -- on the bundled fixture and on reads sampled from the
configs[1] data set against the full 23.5-Mb index.  The preset parameters are chosen to pass this gate, not to hit a
time: a change that makes the kernels faster by moving the spec shows up here as a larger drift."""
import numpy as np
import pytest

from oracle import binding as ob
from telr_amd import synth
from telr_amd._abi import MF_FAITHFUL
from telr_amd.fasta import read_fasta
from telr_amd.presets import preset

# stated thresholds (fractions of the non-secondary records of the sample).  What is left at these settings is the bounded
# look-back: about one read in 600 (sample of 600: 3 of 905 records) spans a spiked insertion next to a reference copy of the same family, more than 128
# anchors sort between the two sides, and the read comes out as primary + supplementary (a split at the insertion)
# where look-back 256 / 5000 joins it into one record with a long I run (3 of ~900 records; both forms carry the
# insertion, tests/test_gpu_stage1_to_loci.py).  Gap fills and extensions contribute nothing on this sample.
MAX_COORD_DRIFT = 0.005      # target / query interval differs
MAX_CORE_DRIFT = 0.005       # a record is missing or has no overlapping counterpart
MAX_SCORE_DRIFT = 0.005      # DP score differs at all
N_SAMPLE = 300


def _records(res):
    out = {}
    for k, a in enumerate(res["alns"]):
        if a["flags"] & 2:
            continue
        out.setdefault(int(a["qid"]), []).append(a)
    return out


def _cigar_digests(res):
    """{(qid, tid, ts, te, qs, qe): hash of the record's CIGAR words} of the non-secondary records"""
    out = {}
    for a, h in zip(res["alns"], res.get("cigar_hash", [])):
        if not a["flags"] & 2:
            out[(int(a["qid"]), int(a["tid"]), int(a["ts"]), int(a["te"]), int(a["qs"]), int(a["qe"]))] = int(h)
    return out


def _map_threads(oix, reads, mo, T=8):
    """the oracle call releases the GIL: T shards of the reads in parallel; query ids restored"""
    from concurrent.futures import ThreadPoolExecutor

    def work(k):
        r = oix.map(reads[k::T], mo)
        al = r["alns"].copy()
        al["qid"] = al["qid"] * T + k
        cg = r["cigars"]
        hs = np.array([hash(cg[a["cigar_off"]:a["cigar_off"] + a["n_cigar"]].tobytes()) for a in al], np.int64)
        return al, hs
    with ThreadPoolExecutor(T) as ex:
        parts = list(ex.map(work, range(T)))
    return {"alns": np.concatenate([p[0] for p in parts]), "cigar_hash": np.concatenate([p[1] for p in parts])}


def drift(oix, reads, mo, parts=4, other=None, forgive_zdrop=False):
    """preset vs the same options with the faithful bits `parts` (4 = all; 0x100 look-back, 0x200 fills, 0x400 extensions),
    or vs the options `other`.  forgive_zdrop: reads whose faithful records contain a z-dropped gap fill (n_ambi > 0 under
    the 0x10000 experiment: minimap2 would break the record there again) are left out of the comparison."""
    mf = mo.copy(); mf.flags |= parts
    if other is not None:
        mf = other
    ra_, rb_ = _map_threads(oix, reads, mo), _map_threads(oix, reads, mf)
    a, b = _records(ra_), _records(rb_)
    ca, cb = _cigar_digests(ra_), _cigar_digests(rb_)
    cigar = sum(1 for k, v in ca.items() if k in cb and cb[k] != v)          # same read, same coordinates, another CIGAR
    n = coord = core = score = 0
    for q in sorted(set(a) | set(b)):
        ra, rb = a.get(q, []), b.get(q, [])
        if forgive_zdrop and any(int(y["n_ambi"]) > 0 for y in rb):
            n += max(len(ra), len(rb))
            continue
        n += max(len(ra), len(rb))
        core += abs(len(ra) - len(rb))
        for x in ra:
            best, ov = None, 0                      # the faithful record of the same read that overlaps x most on the target
            for y in rb:
                if y["tid"] == x["tid"] and (y["flags"] & 8) == (x["flags"] & 8):
                    o = min(int(x["te"]), int(y["te"])) - max(int(x["ts"]), int(y["ts"]))
                    if o > ov:
                        best, ov = y, o
            if best is None:
                core += 1
                continue
            if (int(x["ts"]), int(x["te"]), int(x["qs"]), int(x["qe"])) != (int(best["ts"]), int(best["te"]), int(best["qs"]), int(best["qe"])):
                coord += 1
            if int(x["dp_score"]) != int(best["dp_score"]):
                score += 1
    return dict(n=n, coord=coord / max(1, n), core=core / max(1, n), score=score / max(1, n), cigar=cigar / max(1, n))


@pytest.mark.parametrize("name", ["map-ont", "map-pb", "ngmlr-ont", "ngmlr-pacbio"])
def test_fixture_gap_fills_equal_the_full_band(data_dir, name):
    """REAL reads (the reference's bundled PacBio reads around a jockey insertion, bursty indels): with the preset's
    first-pass band and retry margin no record differs from gap fills over the whole -r band.  (Band factor 8 with the
    touch-only retry rule of round 1 lost two of the 25 map-pb records, factor 4 six.)"""
    _, ts = read_fasta(data_dir + "/ref_38kb.fasta")
    _, qs = read_fasta(data_dir + "/reads.fasta")
    io, mo = preset(name)
    d = drift(ob.OracleIndex(ts, io), qs, mo, parts=0x200)
    assert d["n"] >= 18 and d["core"] == 0 and d["coord"] == 0 and d["score"] == 0, d          # (19 records with the long join: four of the reads across the jockey copy are one record each)
    # what the bounded look-back costs on these reads: nothing
    d = drift(ob.OracleIndex(ts, io), qs, mo, parts=0x100)
    assert d["core"] == 0 and d["coord"] == 0 and d["score"] == 0, d


def test_fixture_extension_cap_is_the_known_difference(data_dir):
    """the end extensions are capped (ext_max 2048 bases in a band of +-31 diagonals): on the fixture two records reach
    ~100 bases further under the faithful mode (reads that run into the jockey copy); everything else is identical"""
    _, ts = read_fasta(data_dir + "/ref_38kb.fasta")
    _, qs = read_fasta(data_dir + "/reads.fasta")
    io, mo = preset("map-pb")
    d = drift(ob.OracleIndex(ts, io), qs, mo, parts=4)
    assert d["core"] == 0 and d["coord"] <= 0.12 and d["score"] <= 0.12, d


@pytest.mark.timeout(1200)
def test_configs1_sample_drift():
    # configs[1]'s genome and read lengths (47 kb), a seventh of its reads: generating all 470 Mbp to map 150 reads was most of this test
    d = synth.make_stage1_dataset(seed=20261002, genome_len=23513712, n_reads=1500, total_bases=70_500_000, read_seed=20261002 + 1000)
    buf, off, ln = d["reads"]
    rng = np.random.default_rng(5)
    n_sample = 150            # (47-kb reads: 7 Mbp through the CPU oracle three times; the 10 x larger gate is tools/faithful_table.py)
    pick = rng.choice(len(ln), size=n_sample, replace=False)
    reads = [bytes(buf[off[i]:off[i] + ln[i]]).decode() for i in pick]
    io, mo = preset("map-ont")
    oix = ob.OracleIndex([bytes(d["ref"]).decode()], io)
    r = drift(oix, reads, mo, parts=4)
    print("configs[1] sample, all bounds lifted:", r)
    assert r["n"] >= n_sample
    one = 1.01 / r["n"]       # one record of this sample
    assert r["core"] <= max(MAX_CORE_DRIFT, one) and r["coord"] <= max(MAX_COORD_DRIFT, one) and r["score"] <= max(MAX_SCORE_DRIFT, one), r
    # the band rule and the extension cap alone: no record moves; at most one in ~450 ends with another DP score (on the
    # 600-read sample: 1 of 905)
    r = drift(oix, reads, mo, parts=0x200 | 0x400)
    print("configs[1] sample, full-band fills + uncapped extensions:", r)
    assert r["core"] == 0 and r["coord"] <= max(0.005, one) and r["score"] <= max(0.005, one), r          # (one of 333 records since the long join: a joined record whose uncapped end extension reaches further)


# ---- sub-read voting (spec 3.10, the ngmlr-* presets): what the candidate search changes against chaining ALL hits ------------
VOTE_MAX_DRIFT = 0.02        # non-secondary records whose coordinates / existence change when the hits are not voted on


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("name,err", [("ngmlr-pacbio", (0.013, 0.065, 0.052)), ("ngmlr-ont", (0.04, 0.02, 0.04))])
def test_subread_voting_against_unpruned_chaining(name, err):
    """NGMLR's candidate search as this engine restates it (256-base sub-reads vote for 32-base diagonal bins, bins with half
    the votes of the best stay) against the same preset chaining every hit: on reads with the preset's error model, sampled from
    a configs[1]-size genome (23.5 Mb, 15 % TE-derived, spiked insertions), primaries and supplementaries must be the same
    records but for a stated fraction.  Secondary records are outside the comparison: dropping the weaker copies of a
    repeat is what the vote is for (NGMLR reports none)."""
    d = synth.make_stage1_dataset(seed=20261002, genome_len=23513712, n_reads=4000, total_bases=36_000_000, err=err, read_seed=20261002 + 77)
    buf, off, ln = d["reads"]
    rng = np.random.default_rng(11)
    pick = rng.choice(len(ln), size=160, replace=False)
    reads = [bytes(buf[off[i]:off[i] + ln[i]]).decode() for i in pick]
    io, mo = preset(name)
    assert mo.vote_len == 256
    plain = mo.copy(); plain.vote_len = 0
    oix = ob.OracleIndex([bytes(d["ref"]).decode()], io)
    r = drift(oix, reads, mo, other=plain)
    print(name, "sub-read voting vs all hits:", r)
    assert r["n"] >= 160
    assert r["core"] <= VOTE_MAX_DRIFT and r["coord"] <= VOTE_MAX_DRIFT, r


# ---- FAITHFUL v2: what the spec leaves out of minimap2 2.22's published behaviour, bit by bit ------------------------------------
# (oracle/telr_oracle.c: MFX_* experiment bits).  tools/faithful_table.py prints the whole table for DESIGN.md; the tests below
# assert the part of it that a CPU run of a few minutes can hold.
BITS = [("look-back 5000", 0x100), ("full-band fills + uncapped extensions", 0x200 | 0x400), ("max_chain_skip scan (look-back 5000, 25 skips)", 0x1000),
        ("high-occurrence seed rescue", 0x2000), ("long join (re-chain with bw_long 20,000)", 0x4000), ("RMQ-style chaining (-r100k, unbounded look-back)", 0x8000)]


def workload(kind, n, hard=False):
    """(index options, map options, reference strings, read strings) of a gate workload; hard: the same workload on the HARD
    genome of telr_amd/synth.py (tandem arrays, microsatellites, low-complexity stretches, segmental duplications, satellite
    blocks next to insertions, reads with error bursts)"""
    hard = True if hard else None
    if kind == "flanks-asm10":
        # 500-base flanks as the liftover cuts them (TELR_liftover.py:167-266), from TE-free and TE-derived reference sequence
        # alike, 0.5 % substitutions (a polished contig), asm10 -N 10
        d = synth.make_stage1_dataset(seed=20261002, genome_len=23513712, n_reads=10, total_bases=100000, hard=hard)
        rng = np.random.default_rng(21)
        ref = d["ref"]
        reads = []
        for _ in range(n):
            p = int(rng.integers(0, len(ref) - 500))
            f = synth.mutate(rng, ref[p:p + 500], 0.005, 0.0, 0.0)
            reads.append(bytes(synth.revcomp_arr(f) if rng.integers(0, 2) else f).decode())
        io, mo = preset("asm10"); mo.best_n = 10
        return io, mo, [bytes(ref).decode()], reads
    if kind in ("clr-map-pb", "clr-ngmlr-pacbio"):
        d = synth.make_stage1_dataset(seed=20261002, genome_len=23513712, n_reads=6000, total_bases=54_000_000, err=(0.013, 0.065, 0.052), read_seed=20261002 + 77, hard=hard)
        io, mo = preset("map-pb" if kind == "clr-map-pb" else "ngmlr-pacbio")
    elif kind == "ont-ngmlr-ont":
        d = synth.make_stage1_dataset(seed=20261002, genome_len=23513712, n_reads=6000, total_bases=54_000_000, read_seed=20261002 + 78, hard=hard)
        io, mo = preset("ngmlr-ont")
    elif kind == "c4-density":
        # the repeat density of configs[4]: a 1,300-family library, 45 % of the sequence TE-derived
        d = synth.make_stage1_dataset(seed=20261002 + 4, genome_len=12_000_000, n_reads=6000, total_bases=54_000_000, n_ins=100, n_fam=1300, te_frac=0.45, gc=0.47, read_seed=20261002 + 79, hard=hard)
        io, mo = preset("map-ont")
    elif kind == "ont-map-ont":
        # the configs[2] default (`--aligner minimap2`, TELR_alignment.py:56-86): round 6, with the hard genome
        d = synth.make_stage1_dataset(seed=20261002, genome_len=23513712, n_reads=6000, total_bases=54_000_000, read_seed=20261002 + 78, hard=hard)
        io, mo = preset("map-ont")
    else:
        raise ValueError(kind)
    buf, off, ln = d["reads"]
    pick = np.random.default_rng(13).choice(len(ln), size=n, replace=False)
    return io, mo, [bytes(d["ref"]).decode()], [bytes(buf[off[i]:off[i] + ln[i]]).decode() for i in pick]


ZDROP_ROWS = ("z-drop 400 (minimap2's value) instead of the preset's", "no z-drop at all (NGMLR has none) instead of the preset's")


def bit_table(kind, n, bits=None, hard=False):
    io, mo, ref, reads = workload(kind, n, hard)
    oix = ob.OracleIndex(ref, io)
    rows = []
    base = _map_threads(oix, reads, mo)
    for name, b in (bits or BITS):
        if b == 0x8000 and mo.bw < 10000:
            continue
        if b == 0x4000 and mo.bw >= 20000:
            continue
        mf = mo.copy(); mf.flags |= b
        r = drift(oix, reads, mo, other=mf)
        rows.append((name, r))
    if kind in ("clr-ngmlr-pacbio", "ont-ngmlr-ont") and bits is None:
        # a13: NGMLR's convex gap cost in exact form (the spec of both presets since round 4) against the two-piece envelope
        # (the preset's q / e / q2 / e2, which apply once cx_scale is 0)
        mf = mo.copy(); mf.cx_scale = 0
        rows.append((CONVEX, drift(oix, reads, mo, other=mf)))
    if mo.zdrop < 400 and bits is None:
        # round 5 set `ngmlr-ont`'s z-drop to 100 (a speed change on the reference's DEFAULT aligner, telr_amd/presets.py): its own rows
        for name, zd in zip(ZDROP_ROWS, (400, 1 << 24)):
            mf = mo.copy(); mf.zdrop = zd
            rows.append((name, drift(oix, reads, mo, other=mf)))
    if mo.bw < 20000:
        mf = mo.copy(); mf.flags |= 0x4000 | 0x10000
        rows.append(("long join, not counting reads whose joined record z-drops in a fill (minimap2 splits it again)", drift(oix, reads, mo, other=mf, forgive_zdrop=True)))
    # z-drop inside fills: records minimap2 would break; MAPQ: decisions of the Sniffles gate (>= 20) that change
    mz = mo.copy(); mz.flags |= 0x10000
    z = _map_threads(oix, reads, mz)["alns"]
    nz = int(((z["flags"] & 2) == 0).sum())
    rows.append(("z-drop inside a gap fill (records minimap2 would split)", dict(n=nz, core=float(((z["n_ambi"] > 0) & ((z["flags"] & 2) == 0)).sum()) / max(1, nz), coord=0.0, score=0.0)))
    mq = mo.copy(); mq.flags |= 0x20000
    a, b2 = base["alns"], _map_threads(oix, reads, mq)["alns"]
    assert len(a) == len(b2)
    ka = {(int(x["qid"]), int(x["tid"]), int(x["ts"]), int(x["te"]), int(x["flags"])): int(x["mapq"]) for x in a if not x["flags"] & 2}
    kb = {(int(x["qid"]), int(x["tid"]), int(x["ts"]), int(x["te"]), int(x["flags"])): int(x["mapq"]) for x in b2 if not x["flags"] & 2}
    flip = sum(1 for k in ka if (ka[k] >= 20) != (kb.get(k, 0) >= 20))
    rows.append(("minimap2's own MAPQ (records whose MAPQ >= 20 decision changes)", dict(n=len(ka), core=flip / max(1, len(ka)), coord=0.0, score=0.0)))
    return rows


CONVEX = "the two-piece envelope instead of NGMLR's convex gap cost (extension 5 -> 1 / 1 -> 0.5, decay 0.15 per gap base; exact = the spec)"
LONG_JOIN = ("long join (re-chain with bw_long 20,000)", "long join, not counting reads whose joined record z-drops in a fill (minimap2 splits it again)")


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("kind,n", [("flanks-asm10", 600), ("clr-map-pb", 300), ("clr-ngmlr-pacbio", 300), ("ont-ngmlr-ont", 300), ("c4-density", 300)])
def test_faithful_v2_bits(kind, n):
    """every omission, on every preset the reference uses: at most 0.5 % of the non-secondary records may change existence or
    coordinates -- except the rows listed in EXPLAINED, whose figures DESIGN.md section 2 states and explains:
      * long join on the `ngmlr-*` presets: NGMLR, the aligner these presets stand for, splits a read at an SV breakpoint by
        design, so they keep bw_long = 0 and a read across a spiked multi-kb insertion (3-6 % of the reads of these samples) is
        primary + supplementary where the minimap2 heuristic would give one record.  (For `map-ont` / `map-pb` the long join IS
        the spec since round 3 -- section 3.11 -- and the row reads 0.00 %: the one-pass chaining within bw_long equals minimap2's
        two rounds with look-back 5000 on every record of the samples.);
    (Round 5: the row "full-band fills + uncapped extensions" of ngmlr-ont is no longer exempt -- its extension band is +-63 and its
    fill band (7, 4) and its z-drop 100 since then: 0.20 % of the records' coordinates on 4,881 records, profiles/r05_faithful_table.md.)
    The row "the two-piece envelope instead of NGMLR's convex gap cost" (a13): the exact form IS the spec of both presets since
    round 4; the row states what the round-3 envelope moved (ngmlr-pacbio 0.18 % of 4,940 records' coordinates, ngmlr-ont 3.2 %:
    listed in EXPLAINED for `ont`)."""
    EXPLAINED = {(k, nm): 0.09 for k in ("clr-ngmlr-pacbio", "ont-ngmlr-ont") for nm in LONG_JOIN}
    EXPLAINED[("ont-ngmlr-ont", CONVEX)] = 0.06          # what made the exact form the spec
    rows = bit_table(kind, n)
    for name, r in rows:
        print("%-16s %-70s n=%4d  records changed %.4f  coordinates %.4f  DP score %.4f" % (kind, name, r["n"], r["core"], r["coord"], r["score"]))
    for name, r in rows:
        # at this sample size ONE record is 0.2-0.3 %: the test lets two records through and leaves the 0.5 % rule itself to the
        # ten times larger run of tools/faithful_table.py (>= 4,600 records per workload: profiles/r04_faithful_table.md)
        lim = max(EXPLAINED.get((kind, name), 0.005), 2.01 / max(1, r["n"]))
        assert r["core"] <= lim and r["coord"] <= lim, (kind, name, r)
        # same coordinates, another CIGAR: bounded like the coordinates for every omission that does not re-score the gaps (the envelope
        # row places small indels differently by design: 8 % / 78 % of the CIGARs, profiles/r05_faithful_table.md)
        if "cigar" in r and name != CONVEX:
            assert r["cigar"] <= lim, (kind, name, r)
