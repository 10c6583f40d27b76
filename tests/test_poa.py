"""Window partial-order consensus (spec 3.13; oracle/telr_oracle.c: tor_poa): hand-made windows and the error reduction on a
simulated contig against the pile-up vote.  CPU only; tests/test_gpu_consensus.py holds HIP against this oracle."""
import numpy as np

from oracle import binding as ob
from telr_amd import synth
from telr_amd.presets import preset
from test_consensus import _aln, M, I, D


def _run(target, reads, recs, min_depth=3):
    alns, cigs = [], []
    for a, c in recs:
        a["cigar_off"] = len(cigs); cigs += c; alns.append(a)
    return ob.consensus(np.concatenate(alns), np.array(cigs, np.uint32), reads, [target], min_depth=min_depth, poa=True)[0]


def test_hand_made_windows():
    rng = np.random.default_rng(5)
    t = bytes(synth.random_seq(rng, 60)).decode()
    # reads that equal the draft: the draft comes back; fewer than min_depth reads: the draft comes back
    assert _run(t, [t] * 3, [_aln(i, 0, [M(60)], 60) for i in range(3)]) == t
    sub = t[:30] + ("A" if t[30] != "A" else "C") + t[31:]
    assert _run(t, [sub] * 2, [_aln(i, 0, [M(60)], 60) for i in range(2)]) == t
    # a substitution carried by three reads beats the draft's base (the draft counts as one sequence)
    assert _run(t, [sub] * 3, [_aln(i, 0, [M(60)], 60) for i in range(3)]) == sub
    # an inserted base and a deleted base, each in three of three reads -- whatever the pairwise CIGAR says about them: the reads
    # are RE-ALIGNED to the graph (here the CIGARs claim plain matches of the wrong length on purpose)
    ins = t[:20] + "G" + t[20:]
    assert _run(t, [ins] * 3, [_aln(i, 0, [M(20), I(1), M(40)], 61) for i in range(3)]) == ins
    dele = t[:40] + t[41:]
    assert _run(t, [dele] * 3, [_aln(i, 0, [M(40), D(1), M(19)], 59) for i in range(3)]) == dele
    # reverse-strand records vote with the reverse complement; secondary / supplementary records do not vote
    rc = bytes(synth.revcomp_arr(np.frombuffer(sub.encode(), np.uint8))).decode()
    assert _run(t, [rc] * 3, [_aln(i, 0, [M(60)], 60, rev=True) for i in range(3)]) == sub
    assert _run(t, [sub] * 3, [_aln(i, 0, [M(60)], 60, flags=f) for i, f in enumerate((1, 2, 4))]) == t
    # a read that lacks more than half of the window (the other allele across an insertion the contig carries) does not vote
    short = t[:10] + t[50:]
    assert _run(t, [short] * 4, [_aln(i, 0, [M(10), D(40), M(10)], 20) for i in range(4)]) == t
    # two windows: a record must cover a window whole to vote there
    t2 = bytes(synth.random_seq(rng, 330)).decode()
    s2 = t2[:250] + ("A" if t2[250] != "A" else "C") + t2[251:]
    got = _run(t2, [s2[190:]] * 3, [_aln(i, 190, [M(140)], 140) for i in range(3)])      # covers only the second window (200..330)
    assert got == s2
    s3 = t2[:100] + ("A" if t2[100] != "A" else "C") + t2[101:]
    assert _run(t2, [s3[50:250]] * 3, [_aln(i, 50, [M(200)], 200) for i in range(3)]) == t2   # covers neither window whole


def test_poa_beats_the_pile_up_on_a_simulated_contig():
    rng = np.random.default_rng(3)
    truth = synth.random_seq(rng, 12000)
    draft = bytes(synth.mutate(rng, truth, 0.005, 0.003, 0.003)).decode()
    reads = []
    for _ in range(40):
        s = int(rng.integers(0, 6000)); r = synth.mutate(rng, truth[s:s + 6000], 0.04, 0.02, 0.04)
        reads.append(bytes(synth.revcomp_arr(r) if rng.integers(0, 2) else r).decode())
    io, mo = preset("map-ont"); mo.bw = 2000
    r = ob.OracleIndex([draft], io).map(reads, mo)
    pile = ob.consensus(r["alns"], r["cigars"], reads, [draft], min_depth=3)[0]
    poa = ob.consensus(r["alns"], r["cigars"], reads, [draft], min_depth=3, poa=True)[0]
    io2, mo2 = preset("asm10")

    def diff(x):
        a = ob.OracleIndex([x], io2).map([bytes(truth).decode()], mo2)["alns"]
        a = a[(a["flags"] & 1) != 0][0]
        return int(a["blen"] - a["mlen"])
    d0, d1, d2 = diff(draft), diff(pile), diff(poa)
    print("differences to the truth: draft %d, pile-up %d, POA %d" % (d0, d1, d2))
    assert d0 >= 100 and d2 <= d0 // 5 and d2 <= d1 + 3, (d0, d1, d2)
