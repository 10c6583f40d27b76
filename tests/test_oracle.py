"""The CPU oracle itself: pinned to the reference's bundled fixture (its only known answer) and
cross-checked against small independent pure-Python restatements."""
import numpy as np
import pytest

from oracle import binding as ob
from telr_amd.presets import preset
from telr_amd.fasta import read_fasta, revcomp
from telr_amd import synth


def _hash64(key, mask):
    key = (~key + (key << 21)) & mask
    key = key ^ key >> 24
    key = ((key + (key << 3)) + (key << 8)) & mask
    key = key ^ key >> 14
    key = ((key + (key << 2)) + (key << 4)) & mask
    key = key ^ key >> 28
    key = (key + (key << 31)) & mask
    return key


def brute_minimizers(seq, k, w):
    """(w,k)-minimizers by the set definition (Li 2018, section 2.1.1), all windows enumerated."""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    mask = (1 << 2 * k) - 1
    ns = len(seq) - k + 1
    xs = []
    for u in range(ns):
        kmer = seq[u:u + k]
        if any(c not in code for c in kmer):
            xs.append(None); continue
        fw = 0
        for c in kmer:
            fw = fw << 2 | code[c]
        rv = 0
        for c in reversed(kmer):
            rv = rv << 2 | (3 - code[c])
        if fw == rv:
            xs.append(None); continue
        z = 0 if fw < rv else 1
        xs.append((_hash64(rv if z else fw, mask) << 8 | k, (u + k - 1) << 1 | z))
    sel = set()
    win = min(w, ns)
    for j in range(0, ns - win + 1):
        vals = [xs[q][0] for q in range(j, j + win) if xs[q] is not None]
        if not vals:
            continue
        m = min(vals)
        for q in range(j, j + win):
            if xs[q] is not None and xs[q][0] == m:
                sel.add(q)
    return [(xs[q][0], xs[q][1]) for q in sorted(sel)]


def brute_minimizers_hpc(seq, k, w):
    """homopolymer-compressed variant: k-mers over runs, span in original bases, position = last base of the last run"""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    runs = []
    i = 0
    while i < len(seq):
        j = i + 1
        while j < len(seq) and (seq[j] == seq[i] or (seq[j] not in code and seq[i] not in code)):
            j += 1
        runs.append((seq[i], i, j - 1)); i = j
    mask = (1 << 2 * k) - 1
    ns = len(runs) - k + 1
    xs = []
    for u in range(ns):
        rr = runs[u:u + k]
        span = rr[-1][2] - rr[0][1] + 1
        if any(c not in code for c, _, _ in rr) or span >= 256:
            xs.append(None); continue
        fw = 0
        for c, _, _ in rr:
            fw = fw << 2 | code[c]
        rv = 0
        for c, _, _ in reversed(rr):
            rv = rv << 2 | (3 - code[c])
        if fw == rv:
            xs.append(None); continue
        z = 0 if fw < rv else 1
        xs.append((_hash64(rv if z else fw, mask) << 8 | span, rr[-1][2] << 1 | z))
    sel = set()
    win = min(w, ns)
    for j in range(0, ns - win + 1):
        vals = [xs[q][0] for q in range(j, j + win) if xs[q] is not None]
        if not vals:
            continue
        m = min(vals)
        for q in range(j, j + win):
            if xs[q] is not None and xs[q][0] == m:
                sel.add(q)
    return [(xs[q][0], xs[q][1]) for q in sorted(sel)]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_hpc_sketch_matches_set_definition(seed):
    rng = np.random.default_rng(50 + seed)
    s = bytes(synth.random_seq(rng, 600)).decode()
    s = s[:100] + "A" * 40 + s[100:300] + "NNNN" + s[300:400] + "T" * 300 + s[400:]     # long homopolymers (span >= 256) and Ns
    x, y = ob.sketch(s, 19, 10, hpc=1)
    want = brute_minimizers_hpc(s, 19, 10)
    assert [(int(a), int(b)) for a, b in zip(x, y)] == want and len(want) > 10


@pytest.mark.parametrize("k,w,n,seed", [(15, 10, 400, 1), (19, 19, 300, 2), (5, 4, 120, 3), (15, 10, 20, 4), (15, 10, 15, 5), (7, 3, 60, 6)])
def test_sketch_matches_set_definition(k, w, n, seed):
    rng = np.random.default_rng(seed)
    s = bytes(synth.random_seq(rng, n)).decode()
    if seed == 3:
        s = s[:40] + "NNN" + s[43:80] + "ACGT" * 5 + s[100:]      # ambiguous bases and a tandem repeat
    x, y = ob.sketch(s, k, w)
    want = brute_minimizers(s, k, w)
    assert [(int(a), int(b)) for a, b in zip(x, y)] == want


def test_sketch_short_and_empty():
    assert len(ob.sketch("", 15, 10)[0]) == 0
    assert len(ob.sketch("ACGTACG", 15, 10)[0]) == 0
    x, _ = ob.sketch("ACGTTGCAAGGCTTA", 15, 10)
    assert len(x) == 1


def simple_two_piece_nw(q, t, mo):
    """full-matrix global alignment score, two-piece affine, plain Python"""
    NEG = -10 ** 9
    m, n = len(q), len(t)
    H = [[NEG] * (n + 1) for _ in range(m + 1)]
    E1 = [[NEG] * (n + 1) for _ in range(m + 1)]; E2 = [[NEG] * (n + 1) for _ in range(m + 1)]
    F1 = [[NEG] * (n + 1) for _ in range(m + 1)]; F2 = [[NEG] * (n + 1) for _ in range(m + 1)]
    H[0][0] = 0
    for j in range(1, n + 1):
        E1[0][j] = -(mo.q + j * mo.e); E2[0][j] = -(mo.q2 + j * mo.e2); H[0][j] = max(E1[0][j], E2[0][j])
    for i in range(1, m + 1):
        F1[i][0] = -(mo.q + i * mo.e); F2[i][0] = -(mo.q2 + i * mo.e2); H[i][0] = max(F1[i][0], F2[i][0])
    for i in range(1, m + 1):
        for j in range(1, n + 1):
            E1[i][j] = max(H[i][j - 1] - mo.q - mo.e, E1[i][j - 1] - mo.e)
            E2[i][j] = max(H[i][j - 1] - mo.q2 - mo.e2, E2[i][j - 1] - mo.e2)
            F1[i][j] = max(H[i - 1][j] - mo.q - mo.e, F1[i - 1][j] - mo.e)
            F2[i][j] = max(H[i - 1][j] - mo.q2 - mo.e2, F2[i - 1][j] - mo.e2)
            s = -mo.sc_ambi if (q[i - 1] == "N" or t[j - 1] == "N") else (mo.a if q[i - 1] == t[j - 1] else -mo.b)
            H[i][j] = max(H[i - 1][j - 1] + s, E1[i][j], E2[i][j], F1[i][j], F2[i][j])
    return H[m][n]


def cigar_score(cig, q, t, mo):
    sc, i, j = 0, 0, 0
    for c in cig:
        op, l = int(c) & 0xf, int(c) >> 4
        if op == 0:
            for x in range(l):
                a, b = q[i + x], t[j + x]
                sc += -mo.sc_ambi if "N" in (a, b) else (mo.a if a == b else -mo.b)
            i += l; j += l
        else:
            sc -= min(mo.q + l * mo.e, mo.q2 + l * mo.e2)
            if op == 1:
                i += l
            else:
                j += l
    assert (i, j) == (len(q), len(t))
    return sc


@pytest.mark.parametrize("seed", range(6))
def test_banded_nw_is_optimal_on_small_inputs(seed):
    """segments shorter than the band are solved exactly: score equals an unbanded reference DP and the
    CIGAR re-scores to the same number"""
    rng = np.random.default_rng(100 + seed)
    _, mo = preset("map-ont" if seed % 2 == 0 else "asm10")
    t = synth.random_seq(rng, int(rng.integers(10, 23)))
    q = synth.mutate(rng, t, 0.1, 0.08, 0.08)
    if len(q) == 0:
        q = t[:3]
    qs, ts = bytes(q).decode(), bytes(t).decode()
    if seed == 4:
        qs = qs[:3] + "N" + qs[4:]
    sc, cig = ob.nw(qs, ts, mo)
    assert sc == simple_two_piece_nw(qs, ts, mo)
    assert cigar_score(cig, qs, ts, mo) == sc


def test_long_gap_uses_second_affine_piece():
    _, mo = preset("map-ont")
    rng = np.random.default_rng(5)
    a = bytes(synth.random_seq(rng, 300)).decode()
    q = a[:150] + a[180:]                        # 30-base deletion: min(4+2*30, 24+30) = 54
    sc, cig = ob.nw(q, a, mo)
    assert sc == 270 * mo.a - 54
    assert ob.cigar_str(cig) == "150M30D120M"


def test_extension_stops_at_divergence():
    _, mo = preset("map-ont")
    rng = np.random.default_rng(6)
    a = bytes(synth.random_seq(rng, 400)).decode(); b = bytes(synth.random_seq(rng, 1500)).decode()
    c = bytes(synth.random_seq(rng, 1500)).decode()
    sc, cig, qe, te = ob.ext(a + b, a + c, mo)
    assert 395 <= qe <= 430 and 395 <= te <= 430 and sc >= 2 * 390
    sc0, cig0, qe0, te0 = ob.ext(b, c, mo)
    assert sc0 < 40 and qe0 < 40


def test_fixture_known_answer(data_dir):
    """The reference's bundled smoke data (test/*.fasta): 13 reads carry a ~4.6 kb jockey copy in minus
    orientation at reference offset ~33,016-33,018; 5 reads span the site."""
    _, ref = read_fasta(data_dir + "/ref_38kb.fasta")
    names, reads = read_fasta(data_dir + "/reads.fasta")
    _, lib = read_fasta(data_dir + "/library.fasta")
    io, mo = preset("map-pb")
    res = ob.OracleIndex(ref, io).map(reads, mo)
    al, cg = res["alns"], res["cigars"]
    # reads that carry the element INSIDE one record (spec 3.11, the long join: minimap2 -r500,20000): an I run of >= 3 kb whose
    # reference position is the site
    joined = set()
    for a in al:
        if a["flags"] & 2:
            continue
        t = int(a["ts"])
        for c in cg[a["cigar_off"]:a["cigar_off"] + a["n_cigar"]]:
            if (int(c) & 15) == 1 and (int(c) >> 4) >= 3000 and abs(t - 33017) <= 40:
                joined.add(int(a["qid"]))
            if (int(c) & 15) != 1:
                t += int(c) >> 4
    # reads broken at the insertion site: an alignment ends / starts within 15 bp of 33017
    broken = set()
    for a in al:
        if a["flags"] & 2:
            continue
        if abs(int(a["te"]) - 33017) <= 15 or abs(int(a["ts"]) - 33017) <= 15:
            if (a["qe"] - a["qs"]) < a["qlen"] - 1000:
                broken.add(int(a["qid"]))
    assert len(broken | joined) >= 13 and len(broken) >= 8 and len(joined) >= 2, (sorted(broken), sorted(joined))
    spanning = {int(a["qid"]) for a in al if a["ts"] < 32500 and a["te"] > 33500 and a["blen"] > 0}
    assert len(spanning) >= 4
    # the unaligned parts of the broken reads are jockey, minus strand relative to the reference
    io2, mo2 = preset("map-pb")
    lix = ob.OracleIndex(lib, io2)
    strands = []
    for q in sorted(broken):
        ra = [a for a in al if a["qid"] == q and (a["flags"] & 1)][0]
        hits = lix.map([reads[q]], mo2)["alns"]
        hits = hits[(hits["flags"] & 1) != 0]
        assert len(hits) == 1 and hits[0]["te"] - hits[0]["ts"] > 500
        te_rev = bool(hits[0]["flags"] & 8); ref_rev = bool(ra["flags"] & 8)
        strands.append("-" if te_rev != ref_rev else "+")
    assert strands.count("-") >= 8 and strands.count("+") == 0


def test_mapping_is_strand_symmetric():
    rng = np.random.default_rng(9)
    g = bytes(synth.random_seq(rng, 30000)).decode()
    r = bytes(synth.mutate(rng, np.frombuffer(g[5000:9000].encode(), np.uint8).copy(), 0.03, 0.01, 0.01)).decode()
    io, mo = preset("map-ont")
    ix = ob.OracleIndex([g], io)
    a = ix.map([r], mo)["alns"]; b = ix.map([revcomp(r)], mo)["alns"]
    assert len(a) == len(b) == 1
    assert (a[0]["flags"] & 8) != (b[0]["flags"] & 8)
    assert abs(int(a[0]["ts"]) - int(b[0]["ts"])) < 30 and abs(int(a[0]["te"]) - int(b[0]["te"])) < 30


def test_depth_medians_oracle():
    from telr_amd._abi import ALN_DTYPE
    al = np.zeros(3, ALN_DTYPE)
    cig = np.array([100 << 4, 10 << 4 | 2, 50 << 4,      # 100M 10D 50M  at 0
                    60 << 4,                             # 60M at 50
                    30 << 4], np.uint32)                 # secondary, ignored
    al["tid"] = 0; al["ts"] = [0, 50, 0]; al["n_cigar"] = [3, 1, 1]; al["cigar_off"] = [0, 3, 4]; al["flags"] = [1, 4, 2]
    m = ob.depth_medians(al, cig, [200], [0, 0, 0, 0], [0, 50, 100, 190], [49, 99, 109, 250])
    assert list(m) == [1.0, 2.0, 1.0, 0.0]


def _hpc_start_case():
    """a target and reads whose first k-mers carry LONGER homopolymer runs than the target's: with homopolymer-compressed
    minimizers (map-pb) the query's span of a k-mer then exceeds the target's, and `end - span + 1` falls before the target"""
    rng = np.random.default_rng(2)
    t = synth.random_seq(rng, 4000)
    # no homopolymer in the first 60 bases of the target
    for i in range(1, 60):
        while t[i] == t[i - 1]:
            t[i] = ord("ACGT"[int(rng.integers(0, 4))])
    reads = []
    for rep in (3, 6, 12):
        q = np.concatenate([np.repeat(t[:40], rep), t[40:2500]])          # every base of the first 40 repeated: same HPC string, longer raw spans
        reads.append(q)
    return [bytes(t).decode()], [bytes(r).decode() for r in reads]


def test_chain_start_is_clamped_at_the_target_start_with_hpc_minimizers():
    """the chain box uses the QUERY minimizer's span for the target side too (as minimap2 does) and clamps the start at 0
    (mm_reg_set_coor): found at configs[3] size by the bundle parity, where the unclamped start read one base before the target"""
    targets, reads = _hpc_start_case()
    io, mo = preset("map-pb")
    r = ob.OracleIndex(targets, io).map(reads, mo)
    assert len(r["alns"]) >= 3
    assert (r["alns"]["ts"] >= 0).all() and (r["alns"]["te"] <= 4000).all() and (r["alns"]["ts"] <= 5).any()
