"""ONE nested insertion (famB inserted inside a reference copy of famA) through the REFERENCE's own glue -- TELR_te.annotate_contig
(intersect -wao, > 10 bp, merge -d 10000 distinct) and TELR_liftover.liftover -- fed with this engine's aligner output for S4, S5
and S7 (tests/golden/nested_locus.json, captured by tools/capture_goldens.py --only-nested from the imported reference):

  * with the long join in the per-locus presets (minimap2 2.22's -r500,20000) the library hit of famA chains ACROSS
    the insertion, overlaps the ALT hit, the reference merges `famA|famB` and its decision tree says "reference";
  * without it the famA hit breaks at the insertion, only famB overlaps the ALT hit, and the verdict is "non-reference" at the
    simulated coordinate.
Both verdicts are the reference's; this host code must give the same two (the 968-vs-914 decision)."""
import json
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nested_locus.json")


def _rnd(n, seed):
    r = random.Random(seed)
    return "".join(r.choice("ACGT") for _ in range(n))


def _locus():
    fl, fr, fam_a, fam_b = _rnd(3000, 101), _rnd(3000, 102), _rnd(2400, 103), _rnd(1500, 104)
    ref = _rnd(20000, 105) + fl + fam_a + fr + _rnd(20000, 106)
    cut = 1100; tsd = fam_a[cut - 6:cut]
    return ref, fl + fam_a[:cut] + fam_b + tsd + fam_a[cut:] + fr, fam_b + tsd, ["famA", "famB"], [fam_a, fam_b], 20000 + 3000 + cut


@pytest.mark.parametrize("label", ["long_join", "no_long_join"])
def test_nested_insertion_gets_the_references_verdict(label):
    from oracle_backend import OracleBackend
    from telr_amd import locus_pipeline, presets as P
    g = json.load(open(GOLD))
    case = g["cases"][label]
    ref, contig, alt, lib_names, lib, pos = _locus()
    assert pos == g["truth_pos"]
    name = g["locus"]
    be = OracleBackend()
    io10, _ = P.preset("asm10")
    ref_te = [["chr2L", str(23000), str(25400), "famA", ".", "+"]]
    with P.override(bw_long=case["bw_long"]):
        res = locus_pipeline.run_loci(be, be.index([ref], io10), ["chr2L"], lambda ch: ref, [dict(name=name, contig=contig, alt=alt, reads=[])], lib_names, lib,
                                      presets="ont", ref_te_rows=ref_te, overlap_af=False)
    assert [list(r[:6]) for r in res["annotation"]] == case["reference_annotation"]
    want = case["reference_liftover_report"]
    assert len(res["liftover"]) == len(want) == 1
    got, exp = res["liftover"][0]["report"], want[0]["report"]
    for k in ("type", "chrom", "start", "end", "family", "strand", "gap", "TSD_length", "comment"):
        assert got.get(k) == exp.get(k), (k, got.get(k), exp.get(k))
    assert (exp["type"] == "reference") == (label == "long_join")
    if label == "no_long_join":
        assert abs(exp["start"] - pos) <= 20 and exp["family"] == "famB"
