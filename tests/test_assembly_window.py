import numpy as np
from telr_amd._abi import ALN_DTYPE
from telr_amd import telr_assembly as ta


def test_breakpoint_bankers_rounding():
    assert ta.breakpoint(10, 11) == 10 and ta.breakpoint(11, 12) == 12 and ta.breakpoint(33017, 33018) == 33018


def test_window_reads_matches_bruteforce():
    rng = np.random.default_rng(4)
    n = 4000
    al = np.zeros(n, ALN_DTYPE)
    al["qid"] = rng.integers(0, 600, n); al["tid"] = rng.integers(0, 3, n)
    al["ts"] = rng.integers(0, 200000, n); al["te"] = al["ts"] + rng.integers(50, 15000, n)
    al["flags"] = rng.choice([1, 2, 4], n)
    loci = [["chrB", str(p), str(p + int(rng.integers(0, 30)))] for p in rng.integers(0, 210000, 60)] + [["chrZ", "5", "6"], ["chrA", "200", "300"]]
    ids = {"chrA": 0, "chrB": 1, "chrC": 2}
    got = ta.window_reads(al, ids, loci)
    for row, g in zip(loci, got):
        c = ids.get(row[0], -1); bp = ta.breakpoint(row[1], row[2]); s, e = max(0, bp - 1000), bp + 1000
        want = sorted({int(a["qid"]) for a in al if a["tid"] == c and a["ts"] < e and a["te"] > s})
        assert g.tolist() == want
    rows = ta.annotate_vcf_with_counts(loci, got)
    assert rows[0][-1] == str(len(got[0]))
