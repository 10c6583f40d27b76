"""HIP pile-up consensus (telr_consensus_build: k_pile_count / k_pile_call) against the oracle's (tor_consensus), string for
string, on simulated loci: reads of both strands with ONT-like errors, reads with N bases, a contig no read maps to, a contig
with lower-case and N draft bases; and the functional check that polishing moves the drafts towards the truth."""
import numpy as np
import pytest

from oracle import binding as ob
from telr_amd import synth, telr_assembly
from telr_amd.presets import preset

pytestmark = pytest.mark.gpu


def _loci(seed, n_loci=12, depth=30):
    rng = np.random.default_rng(seed)
    truths, drafts, reads = [], [], []
    for k in range(n_loci):
        L = int(rng.integers(6000, 15000))
        truth = synth.random_seq(rng, L)
        draft = synth.mutate(rng, truth, 0.005, 0.003, 0.003)
        if k == 3:
            draft = draft.copy(); draft[100:140] |= 32; draft[500:503] = ord("N")
        rs = []
        for _ in range(depth if k != 5 else 0):          # locus 5: no reads at all
            s = int(rng.integers(0, max(1, L - 4000))); r = synth.mutate(rng, truth[s:s + int(rng.integers(2000, 6000))], 0.04, 0.02, 0.04)
            if rng.random() < 0.1:
                r = r.copy(); r[10:14] = ord("N")
            rs.append(bytes(synth.revcomp_arr(r) if rng.integers(0, 2) else r).decode())
        truths.append(bytes(truth).decode()); drafts.append(bytes(draft).decode()); reads.append(rs)
    return truths, drafts, reads


@pytest.mark.parametrize("seed,pname", [(1, "map-ont"), (2, "map-pb")])
def test_hip_consensus_equals_oracle(engine, seed, pname):
    truths, drafts, reads = _loci(seed)
    io, mo = preset(pname); mo.bw = 2000
    qt = np.array([k for k, rs in enumerate(reads) for _ in rs], np.int32)
    flat = [r for rs in reads for r in rs]
    ix = engine.index(drafts, io)
    qset = engine.seqset(flat)
    r = ix.map_raw(qset, mo, qtarget=qt)
    try:
        res = ix.result_arrays(r)
        for md in (3, 1, 8):
            got = ix.consensus(r, qset, min_depth=md)
            want = ob.consensus(res.alns, res.cigars, flat, drafts, min_depth=md)
            assert got == want, [i for i in range(len(got)) if got[i] != want[i]]
    finally:
        ix.free_raw(r)
    assert got[5] == drafts[5]                                     # nothing mapped: the draft stays
    assert got[3] != drafts[3] and got[3].upper() == got[3]


def test_polish_consensus_moves_the_drafts_to_the_truth(engine):
    truths, drafts, reads = _loci(7, n_loci=8, depth=40)
    out = telr_assembly.polish_consensus(engine, ["c%d" % i for i in range(8)], drafts, reads, presets="ont", iterations=2)
    io2, mo2 = preset("asm10")

    def diffs(seqs):
        tot = 0
        for k, s in enumerate(seqs):
            if k == 5:
                continue
            a = engine.index([s], io2).map([truths[k]], mo2).alns
            a = a[(a["flags"] & 1) != 0][0]
            tot += int(a["blen"] - a["mlen"])
        return tot
    d0, d1 = diffs(drafts), diffs(out)
    assert d0 >= 400 and d1 <= d0 // 5, (d0, d1)
    assert out[5] == drafts[5]


@pytest.mark.parametrize("seed,pname", [(1, "map-ont"), (2, "map-pb")])
def test_hip_poa_equals_oracle(engine, seed, pname):
    """the window partial-order consensus (telr_poa_build: k_poa_window, one wave per window) against the oracle's tor_poa,
    string for string: the same pieces, the same graphs, the same heaviest-bundle paths"""
    truths, drafts, reads = _loci(seed, n_loci=6, depth=24)
    io, mo = preset(pname); mo.bw = 2000
    qt = np.array([k for k, rs in enumerate(reads) for _ in rs], np.int32)
    flat = [r for rs in reads for r in rs]
    ix = engine.index(drafts, io)
    qset = engine.seqset(flat)
    r = ix.map_raw(qset, mo, qtarget=qt)
    try:
        res = ix.result_arrays(r)
        for md in (3, 1, 12):
            got = ix.consensus(r, qset, min_depth=md, poa=True)
            want = ob.consensus(res.alns, res.cigars, flat, drafts, min_depth=md, poa=True)
            assert got == want, [(i, len(got[i]), len(want[i])) for i in range(len(got)) if got[i] != want[i]]
    finally:
        ix.free_raw(r)
    assert got[5] == drafts[5].upper().replace("R", "N") or got[5] == drafts[5]        # nothing mapped: the draft stays
    assert got[3].upper() == got[3]


def test_polish_poa_against_the_pile_up(engine):
    truths, drafts, reads = _loci(7, n_loci=8, depth=40)
    names = ["c%d" % i for i in range(8)]
    pile = telr_assembly.polish_consensus(engine, names, drafts, reads, presets="ont", iterations=1)
    poa = telr_assembly.polish_consensus(engine, names, drafts, reads, presets="ont", iterations=1, method="poa")
    io2, mo2 = preset("asm10")

    def diffs(seqs):
        tot = 0
        for k, s in enumerate(seqs):
            if k == 5:
                continue
            a = engine.index([s], io2).map([truths[k]], mo2).alns
            a = a[(a["flags"] & 1) != 0][0]
            tot += int(a["blen"] - a["mlen"])
        return tot
    d0, d1, d2 = diffs(drafts), diffs(pile), diffs(poa)
    print("differences to the truth over 7 contigs: drafts %d, pile-up %d, POA %d" % (d0, d1, d2))
    assert d0 >= 400 and d2 <= d0 // 5 and d2 <= d1 + 10, (d0, d1, d2)
    assert poa[5] == drafts[5]


def test_hip_poa_equals_oracle_on_two_allele_loci(engine):
    """the same comparison where the reads come from TWO alleles (tests/locus_data.py: half of the reads lack the element the
    contig carries and cross it with one long D): pieces under the long-indel rule, windows at the junctions"""
    from locus_data import make_loci
    ref, lib_names, lib, loci, truth = make_loci()
    io, mo = preset("map-ont"); mo.bw = 2000
    drafts = [l["contig"] for l in loci]
    qt = np.array([k for k, l in enumerate(loci) for _ in l["reads"]], np.int32)
    flat = [r for l in loci for r in l["reads"]]
    ix = engine.index(drafts, io)
    qset = engine.seqset(flat)
    r = ix.map_raw(qset, mo, qtarget=qt)
    try:
        res = ix.result_arrays(r)
        got = ix.consensus(r, qset, min_depth=3, poa=True)
        want = ob.consensus(res.alns, res.cigars, flat, drafts, min_depth=3, poa=True)
        bad = [(i, len(got[i]), len(want[i]), next((p for p in range(min(len(got[i]), len(want[i]))) if got[i][p] != want[i][p]), -1)) for i in range(len(got)) if got[i] != want[i]]
        assert not bad, bad
    finally:
        ix.free_raw(r)


def test_hip_poa_equals_oracle_with_long_indels_inside_the_windows(engine):
    """reads that carry 18-28-base deletions and insertions against the draft (under the 30-base rule their pieces vote): edges
    of the window graphs that span more ranks than the kernel's LDS ring of rows holds (POA_RING = 16) -- the sweep then writes
    its rows to the slot and reads such a predecessor from there -- and chains of inserted nodes behind one column"""
    rng = np.random.default_rng(20261003)
    drafts, reads = [], []
    for k in range(5):
        L = int(rng.integers(5000, 9000))
        truth = synth.random_seq(rng, L)
        drafts.append(bytes(synth.mutate(rng, truth, 0.004, 0.002, 0.002)).decode())
        # the SAME long differences in a third of the reads each (alleles), so that they are heavy enough to stay in the graphs
        events = sorted(int(x) for x in rng.integers(300, L - 300, 6))
        kinds = [(int(rng.integers(0, 2)), int(rng.integers(18, 29)), synth.random_seq(rng, 28)) for _ in events]
        rs = []
        for i in range(36):
            t = truth
            if i % 3:
                parts, last = [], 0
                for (pos, (ins, ln, filler)) in zip(events, kinds):
                    if (i + pos) % 2:
                        continue
                    parts.append(t[last:pos])
                    if ins:
                        parts.append(filler[:ln]); last = pos
                    else:
                        last = pos + ln
                parts.append(t[last:])
                t = np.concatenate(parts)
            s = int(rng.integers(0, max(1, len(t) - 3500))); r = synth.mutate(rng, t[s:s + int(rng.integers(2500, 5000))], 0.03, 0.015, 0.02)
            rs.append(bytes(synth.revcomp_arr(r) if rng.integers(0, 2) else r).decode())
        reads.append(rs)
    io, mo = preset("map-ont"); mo.bw = 2000
    qt = np.array([k for k, rs in enumerate(reads) for _ in rs], np.int32)
    flat = [r for rs in reads for r in rs]
    ix = engine.index(drafts, io)
    qset = engine.seqset(flat)
    r = ix.map_raw(qset, mo, qtarget=qt)
    try:
        res = ix.result_arrays(r)
        # the long differences are in the records (else this test tests nothing)
        ops = res.cigars & 0xf; lens = res.cigars >> 4
        assert int(((ops == 2) & (lens >= 18) & (lens <= 30)).sum()) >= 20 and int(((ops == 1) & (lens >= 18) & (lens <= 30)).sum()) >= 20
        for md in (3, 1):
            got = ix.consensus(r, qset, min_depth=md, poa=True)
            want = ob.consensus(res.alns, res.cigars, flat, drafts, min_depth=md, poa=True)
            assert got == want, [(i, len(got[i]), len(want[i])) for i in range(len(got)) if got[i] != want[i]]
    finally:
        ix.free_raw(r)


@pytest.mark.parametrize("seed", [31, 32])
def test_hip_poa_equals_oracle_on_random_shapes(engine, seed):
    """tests/fuzz_poa.py: contigs shorter than a window / a base past a border, depths 0 to 80 (beyond the 64-piece cap), clean to
    noisy reads, N runs, 12-45-base indels either side of the 30-base rule, both presets, min_depth 3 and 1"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_poa
    fuzz_poa.run(engine, 12, seed)


def test_hip_poa_equals_oracle_where_reads_and_draft_carry_n_in_the_same_columns(engine):
    """an N of the draft (node base 4) under an N of a read.  The oracle scores N against N as a mismatch
    (`seq[j-1] == base && seq[j-1] < 4`); the sweep's copy of a node's base is 6 for an ambiguous node, so the kernel does not
    depend on the rule that keeps pieces with an N from voting to agree.  Reads without an N cross the draft's Ns
    (they vote against an N node), reads with an N in the same columns do not vote; both sides must give the same strings."""
    rng = np.random.default_rng(77)
    drafts, reads = [], []
    for k in range(3):
        L = 700 + 150 * k
        truth = synth.random_seq(rng, L)
        draft = synth.mutate(rng, truth, 0.01, 0.004, 0.004).copy()
        for p in (60, 61, 62, 250, 410, 411):
            draft[p] = ord("N")
        rs = []
        for x in range(24):
            r = synth.mutate(rng, truth, 0.04, 0.02, 0.03).copy()
            if x % 3 == 0:                                  # an N about where the draft has one, and one somewhere else
                r[61] = ord("N"); r[int(rng.integers(300, len(r) - 10))] = ord("N")
            rs.append(bytes(synth.revcomp_arr(r) if x & 1 else r).decode())
        drafts.append(bytes(draft).decode()); reads.append(rs)
    io, mo = preset("map-ont"); mo.bw = 2000
    flat = [r for rs in reads for r in rs]
    qt = np.array([k for k, rs in enumerate(reads) for _ in rs], np.int32)
    ix = engine.index(drafts, io)
    qset = engine.seqset(flat)
    r = ix.map_raw(qset, mo, qtarget=qt)
    try:
        res = ix.result_arrays(r)
        for md in (3, 1):
            got = ix.consensus(r, qset, min_depth=md, poa=True)
            want = ob.consensus(res.alns, res.cigars, flat, drafts, min_depth=md, poa=True)
            assert got == want, [(i, len(got[i]), len(want[i])) for i in range(len(got)) if got[i] != want[i]]
    finally:
        ix.free_raw(r); ix.free(); qset.free()


@pytest.mark.parametrize("method", ["pileup", "poa"])
def test_polish_in_two_halves_equals_one_call(engine, method, monkeypatch):
    """round 6: from 64 loci on `polish_consensus` runs the loci as two halves at a time, the second on the engine's second context in a host
    thread (the host work of one half under the other's kernels): per-locus results do not depend on what else is in the call, so the strings
    equal those of ONE call (TELR_POLISH_HALVES=1), two iterations included; a locus without reads and one with N / lower-case draft bases in both halves"""
    truths, drafts, reads = _loci(11, n_loci=72, depth=12)
    reads[40] = []; d = np.frombuffer(drafts[50].encode(), np.uint8).copy(); d[200:230] |= 32; d[700:702] = ord("N"); drafts[50] = bytes(d).decode()
    names = ["c%d" % i for i in range(len(drafts))]
    monkeypatch.setenv("TELR_POLISH_HALVES", "1")
    one = telr_assembly.polish_consensus(engine, names, drafts, reads, presets="ont", iterations=2, method=method)
    monkeypatch.delenv("TELR_POLISH_HALVES")
    t = {}
    two = telr_assembly.polish_consensus(engine, names, drafts, reads, presets="ont", iterations=2, method=method, timings=t)
    assert "second_half_s" in t                      # the two-halves path ran
    assert len(one) == len(two) == 72
    for k in range(72):
        assert one[k] == two[k], "locus %d" % k
    assert sum(a != b for a, b in zip(one, drafts)) >= 60        # and polishing did something
