from telr_amd import intervals as iv


def test_sort_and_closest_same_strand():
    a = [["chr2", "100", "200", "a1", "60", "+"], ["chr1", "500", "600", "a2", "60", "-"], ["chr1", "100", "200", "a3", "3", "+"]]
    assert [r[3] for r in iv.bed_sort(a)] == ["a3", "a2", "a1"]
    b = [["chr1", "150", "250", "b1", "60", "+"], ["chr1", "200", "300", "b2", "60", "+"], ["chr1", "700", "800", "b3", "60", "-"],
         ["chr1", "300", "400", "b4", "60", "-"]]
    out = iv.closest_same_strand(iv.bed_sort(a), b)
    # a3 (+, 100-200): overlaps b1 -> 0 ; b2 book-ended would be 1, not reported
    assert [(r[3], r[9], r[12]) for r in out if r[3] == "a3"] == [("a3", "b1", "0")]
    # a2 (-, 500-600): b3 at 700 -> 101, b4 ends 400 -> 101 : tie, both reported in B order
    assert [(r[9], r[12]) for r in out if r[3] == "a2"] == [("b3", "101"), ("b4", "101")]
    # a1 on chr2: nothing -> filler row
    assert [r[6:] for r in out if r[3] == "a1"] == [[".", "-1", "-1", ".", "-1", ".", "-1"]]


def test_closest_signed_k():
    a = [["chr1", "1000", "1100", "q", ".", "+"]]
    b = [["chr1", "100", "200", "te1", ".", "+"], ["chr1", "1100", "1200", "te2", ".", "+"], ["chr1", "1050", "1060", "te3", ".", "-"],
         ["chr1", "5000", "6000", "te4", ".", "+"], ["chr2", "1000", "1100", "te5", ".", "+"]]
    out = iv.closest_signed_k(a, b, k=2)
    assert [(r[9], r[12]) for r in out] == [("te3", "0"), ("te2", "1")]
    out = iv.closest_signed_k(a, b, k=5)
    assert [(r[9], r[12]) for r in out] == [("te3", "0"), ("te2", "1"), ("te1", "-801"), ("te4", "3901")]


def test_merge_collapse():
    rows = iv.bed_sort([["c", "10", "20", "x"], ["c", "20", "30", "y"], ["c", "31", "40", "z"], ["d", "1", "2", "w"]])
    assert iv.merge_collapse(rows) == [["c", "10", "30", "x,y"], ["c", "31", "40", "z"], ["d", "1", "2", "w"]]
