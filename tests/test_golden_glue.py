"""Host glue vs golden vectors captured from the reference's own Python (tools/capture_goldens.py).

These pin the reference-side semantics of the path: the liftover decision tree and its quirks
(TELR_liftover.py:393-937, 1062-1141) and the allele-frequency arithmetic (TELR_te.py:518-575, 757-884).
"""
import json
import os
import random

import pytest

from telr_amd import telr_liftover as tl
from telr_amd import telr_af as af

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def _rnd_seq(n, seed):
    r = random.Random(seed)
    return "".join(r.choice("ACGT") for _ in range(n))


LS = _load("liftover_single.json")


@pytest.mark.parametrize("case", LS["cases"], ids=[c["name"] for c in LS["cases"]])
def test_liftover_single_annotation(case):
    ref = {k: _rnd_seq(*v) for k, v in LS["ref_seed"].items()}
    a = case["annotation"]
    f5, f3 = tl.flank_intervals(a["start"], a["end"], 500, case["contig_length"])
    hits5 = None if f5 is None else [tl.parse_paf_line(l) for l in case["paf"].get("5p", [])]
    hits3 = None if f3 is None else [tl.parse_paf_line(l) for l in case["paf"].get("3p", [])]
    got = tl.lift_annotation(a["chrom"], a["start"], a["end"], a["family"], a["strand"], hits5, hits3, case["ref_te_bed"],
                             lambda ch: ref[ch], 500, case["flank_gap_max"], case["flank_overlap_max"])
    assert got == case["expected"]


def test_liftover_driver_dedup_and_reports(tmp_path):
    g = _load("liftover_driver.json")
    ref = {k: _rnd_seq(*v) for k, v in g["ref_seed"].items()}
    paf = {tuple(k.split("|")): [tl.parse_paf_line(l) for l in v] for k, v in g["paf"].items()}

    def mapper(queries, qnames):
        out = {}
        for qi, qn in enumerate(qnames):
            contig, rng = qn.rsplit(":", 1)
            s, e = (int(x) for x in rng.split("-"))
            for (prefix, side), hits in paf.items():
                if hits and hits[0].qname == qn:
                    out[qi] = hits
            assert len(queries[qi]) == e - s
        return out
    data, summ = tl.liftover(mapper, g["contig_seqs"], g["bed1"], lambda ch: ref[ch], None, 500, 20, 20, out_dir=str(tmp_path))
    assert data == g["expected_report"]
    assert summ == g["expected_summary"]
    assert (tmp_path / "liftover_nonref.bed").read_text() == g["expected_nonref_bed"]
    assert json.loads((tmp_path / "liftover_report.json").read_text()) == g["expected_report"]


AF = _load("af.json")


@pytest.mark.parametrize("case", AF["cases"], ids=[c["name"] for c in AF["cases"]])
def test_allele_frequency(case):
    s, e = case["te"]
    fi, fo, ti, to = case["params"]
    ivs = af.locus_intervals(s, e, case["contig_length"], fi, fo, ti, to)
    # the depth queries the reference issues, in its order: fw te5, te3, flank5, flank3, then rc
    want = [tuple(x) for x in case["requested_intervals"]]
    got = [(tag, x[0], x[1]) for tag in ("fw", "rc") for x in ivs[tag] if x is not None]
    assert got == want
    meds = {"fw": [None] * 4, "rc": [None] * 4}
    for tag, m in (("fw", case["medians_fw"]), ("rc", case["medians_rc"])):
        for k, x in enumerate(ivs[tag]):
            meds[tag][k] = None if x is None else m[k]
    assert af.freq_table(meds) == case["expected"]


def test_pure_helpers():
    g = _load("helpers.json")
    for args, want in g["get_coord"]:
        assert list(tl.get_coord(*args)) == want
    for args, want in g["absmin"]:
        assert tl.absmin(*args) == want
    for args, want in g["choose_new_size"]:
        assert tl.choose_new_size(*args) == want
    for args, want in g["check_nums_similar"]:
        assert tl.check_nums_similar(*args) == want
    for args, want in g["get_te_flank_ratio"]:
        assert af.get_te_flank_ratio(*args) == want


def test_depth_region_semantics():
    # samtools region chr:S-E is 1-based inclusive; the reference feeds 0-based numbers (TELR_te.py:870-884)
    assert af.depth_region(3050, 3100) == (3049, 3099)      # 51 positions
    assert af.depth_region(0, 50) == (0, 49)
    assert af.depth_region(2700, 2800) == (2699, 2799)      # 101 positions


def test_locus_intervals_batch_equals_the_scalar_functions():
    """the vectorised layout of the 8 depth queries per locus (telr_af.locus_intervals_batch) = locus_intervals, locus by locus,
    over random TE coordinates incl. TEs at the contig ends, TEs shorter than offset + interval, and interval size 0"""
    import numpy as np
    from telr_amd import telr_af
    rng = np.random.default_rng(11)
    for params in ((100, 200, 50, 50), (100, 200, 0, 50), (30, 10, 500, 5), (1, 0, 1, 0)):
        L = rng.integers(50, 5000, size=400)
        s = (rng.random(400) * L).astype(np.int64)
        e = np.minimum(L, s + rng.integers(1, 1500, size=400))
        s[:5] = 0; e[5:10] = L[5:10]
        lo, hi, valid = telr_af.locus_intervals_batch(s, e, L, *params)
        for k in range(400):
            want = telr_af.locus_intervals(int(s[k]), int(e[k]), int(L[k]), *params)
            for o, tag in enumerate(("fw", "rc")):
                for j, x in enumerate(want[tag]):
                    if x is None:
                        assert not valid[k, o, j]
                    else:
                        assert valid[k, o, j] and (int(lo[k, o, j]), int(hi[k, o, j])) == x, (params, k, tag, j)
