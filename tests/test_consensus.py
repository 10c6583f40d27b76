"""Pile-up consensus (spec 3.12; oracle/telr_oracle.c: tor_consensus): the voting rules on hand-made pile-ups and the error
reduction on a simulated locus.  CPU only; tests/test_gpu_consensus.py holds HIP against this oracle."""
import numpy as np
import pytest

from oracle import binding as ob
from telr_amd import synth
from telr_amd._abi import ALN_DTYPE
from telr_amd.presets import preset


def _aln(qid, ts, cig, qlen, rev=False, flags=1, qs=0):
    a = np.zeros(1, ALN_DTYPE)
    a["qid"] = qid; a["tid"] = 0; a["qlen"] = qlen; a["ts"] = ts; a["flags"] = flags | (8 if rev else 0)
    a["qs"] = qs; a["qe"] = qs + sum(c >> 4 for c in cig if (c & 15) in (0, 1)); a["te"] = ts + sum(c >> 4 for c in cig if (c & 15) in (0, 2))
    a["n_cigar"] = len(cig)
    return a, cig


def _run(target, reads, recs, min_depth=3):
    alns, cigs = [], []
    for a, c in recs:
        a["cigar_off"] = len(cigs); cigs += c; alns.append(a)
    return ob.consensus(np.concatenate(alns), np.array(cigs, np.uint32), reads, [target], min_depth=min_depth)[0]


M, I, D = (lambda n: n << 4), (lambda n: n << 4 | 1), (lambda n: n << 4 | 2)


def test_voting_rules():
    t = "ACGTACGTAC"
    # three reads agree on a substitution at position 4 (A -> G): the majority wins; flanks below min_depth keep the draft
    reads = ["CGTGCGT"] * 3
    assert _run(t, reads, [_aln(i, 1, [M(7)], 7) for i in range(3)]) == "ACGTGCGTAC"
    # two of three is still a majority; one of three is not
    assert _run(t, ["CGTGCGT", "CGTGCGT", "CGTACGT"], [_aln(i, 1, [M(7)], 7) for i in range(3)]) == "ACGTGCGTAC"
    assert _run(t, ["CGTGCGT", "CGTACGT", "CGTACGT"], [_aln(i, 1, [M(7)], 7) for i in range(3)]) == "ACGTACGTAC"
    # a tie between the draft base and another base keeps the draft (2 x A, 2 x G at position 4)
    assert _run(t, ["CGTGCGT", "CGTGCGT", "CGTACGT", "CGTACGT"], [_aln(i, 1, [M(7)], 7) for i in range(4)]) == "ACGTACGTAC"
    # deletion: dropped only with a strict majority (2 of 3 yes, 2 of 4 no)
    rd = ["CGTCGT", "CGTCGT", "CGTACGT"]
    assert _run(t, rd, [_aln(0, 1, [M(3), D(1), M(3)], 6), _aln(1, 1, [M(3), D(1), M(3)], 6), _aln(2, 1, [M(7)], 7)]) == "ACGTCGTAC"
    rd4 = rd + ["CGTACGT"]
    assert _run(t, rd4, [_aln(0, 1, [M(3), D(1), M(3)], 6), _aln(1, 1, [M(3), D(1), M(3)], 6), _aln(2, 1, [M(7)], 7), _aln(3, 1, [M(7)], 7)]) == "ACGTACGTAC"
    # insertion after position 3: two of three reads insert "GG", the third "G": column 0 has 3 votes, column 1 has 2 of cov 3 -> both kept
    ri = ["CGTGGACGT", "CGTGGACGT", "CGTGACGT"]
    recs = [_aln(0, 1, [M(3), I(2), M(4)], 9), _aln(1, 1, [M(3), I(2), M(4)], 9), _aln(2, 1, [M(3), I(1), M(4)], 8)]
    assert _run(t, ri, recs) == "ACGTGGACGTAC"
    # secondary and supplementary records do not vote; min_depth keeps the draft
    recs = [_aln(i, 1, [M(7)], 7, flags=f) for i, f in enumerate((1, 2, 4))]
    assert _run(t, ["CGTGCGT"] * 3, recs) == t
    assert _run(t, ["CGTGCGT"] * 2, [_aln(i, 1, [M(7)], 7) for i in range(2)], min_depth=3) == t
    assert _run(t, ["CGTGCGT"] * 2, [_aln(i, 1, [M(7)], 7) for i in range(2)], min_depth=2) == "ACGTGCGTAC"
    # a reverse-strand record votes with the reverse complement of its read; an ambiguous read base counts for coverage only
    assert _run(t, ["ACGCACG"] * 3, [_aln(i, 1, [M(7)], 7, rev=True) for i in range(3)]) == "ACGTGCGTAC"
    assert _run(t, ["CGTNCGT"] * 3, [_aln(i, 1, [M(7)], 7) for i in range(3)]) == t
    # a long D (longer than 30: a read of the other allele across an insertion the contig carries) does not vote: the bases stay
    tl = "ACGT" * 20
    long_d = [_aln(i, 2, [M(10), D(40), M(10)], 20) for i in range(3)]
    assert _run(tl, [tl[2:12] + tl[52:62]] * 3, long_d) == tl
    short_d = [_aln(i, 2, [M(10), D(4), M(10)], 20) for i in range(3)]
    assert _run(tl, [tl[2:12] + tl[16:26]] * 3, short_d) == tl[:12] + tl[16:]
    # lower-case / IUPAC draft bases come out as the engine sees them
    assert _run("acgtRcgtac", ["CGT"] * 3, [_aln(i, 1, [M(3)], 3) for i in range(3)]) == "ACGTNCGTAC"


def test_polishing_reduces_the_error_of_a_simulated_contig():
    rng = np.random.default_rng(3)
    truth = synth.random_seq(rng, 12000)
    draft = bytes(synth.mutate(rng, truth, 0.005, 0.003, 0.003)).decode()
    reads = []
    for _ in range(40):
        s = int(rng.integers(0, 6000)); r = synth.mutate(rng, truth[s:s + 6000], 0.04, 0.02, 0.04)
        reads.append(bytes(synth.revcomp_arr(r) if rng.integers(0, 2) else r).decode())
    io, mo = preset("map-ont"); mo.bw = 2000
    r = ob.OracleIndex([draft], io).map(reads, mo)
    cons = ob.consensus(r["alns"], r["cigars"], reads, [draft], min_depth=3)[0]
    io2, mo2 = preset("asm10")

    def diff(x):
        a = ob.OracleIndex([x], io2).map([bytes(truth).decode()], mo2)["alns"]
        a = a[(a["flags"] & 1) != 0][0]
        return int(a["blen"] - a["mlen"])
    d0, d1 = diff(draft), diff(cons)
    assert d0 >= 100 and d1 <= d0 // 5, (d0, d1)
