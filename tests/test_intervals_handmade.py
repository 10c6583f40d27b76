"""telr_amd/intervals.py against cases derived BY HAND from the bedtools 2.30 documentation (tests/golden/bedtools_handmade.json;
each case carries its derivation).  Unlike the captured goldens, which answer the reference's bedtools calls WITH intervals.py,
these pin the closest / merge / intersect / sort rules independently of the implementation."""
import json
import os

import pytest

from telr_amd import intervals as iv

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bedtools_handmade.json")))


@pytest.mark.parametrize("case", G["cases"], ids=[c["tool"] + ":" + c["name"] for c in G["cases"]])
def test_case(case):
    a, b, args = case["a"], case["b"], case["args"]
    if case["tool"] == "closest_s_d_tall":
        got = iv.closest_same_strand(a, b)
    elif case["tool"] == "closest_D_ref_k":
        got = iv.closest_signed_k(a, b, k=args["k"])
    elif case["tool"] == "merge_distinct":
        got = iv.merge_distinct(a, args["d"], args["cols"], args["delim"])
    elif case["tool"] == "intersect_wao":
        got = iv.intersect_wao(a, b)
    else:
        got = iv.bed_sort(a)
    assert [list(map(str, r)) for r in got] == case["expected"], case["derivation"]
