"""Window-read selection (stage 1 -> per-locus hand-off, a12) against a golden captured from the reference's own
`prep_assembly_inputs(read_type="all")` (src/telr/TELR_assembly.py:384-462) run with pysam / seqtk / Bio stubbed
(tools/capture_goldens.py: capture_prep_assembly): the read set per locus and the `.new` copy of the locus table."""
import json
import os

import numpy as np

from telr_amd import telr_assembly as ta
from telr_amd._abi import ALN_DTYPE

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prep_assembly.json")))


def _records():
    names = sorted({r[0] for r in G["records"]})
    qid = {n: i for i, n in enumerate(names)}
    chrom_ids = {c: i for i, c in enumerate(sorted(G["chrom_len"]))}
    mapped = [r for r in G["records"] if not (r[4] & 4)]
    al = np.zeros(len(mapped), ALN_DTYPE)
    for k, (n, c, s, e, fl) in enumerate(mapped):
        al[k]["qid"] = qid[n]; al[k]["tid"] = chrom_ids[c]; al[k]["ts"] = s; al[k]["te"] = e
        al[k]["flags"] = (2 if fl & 256 else 4 if fl & 2048 else 1) | (8 if fl & 16 else 0)
    return names, chrom_ids, al


def test_window_reads_equal_the_reference():
    names, chrom_ids, al = _records()
    got = ta.window_reads(al, chrom_ids, G["vcf_rows"])
    assert [sorted(names[i] for i in x) for x in got] == G["expected_reads_per_locus"]
    # secondary-only and supplementary records select their read, abutting records do not (htslib region rule)
    first = {names[i] for i in got[0]}
    assert "edge_secondary_only" in first and "edge_end_past_start" in first and "edge_start_before_end" in first
    assert "edge_end_at_start" not in first and "edge_start_at_end" not in first and "edge_unmapped" not in first


def test_new_table_text():
    names, chrom_ids, al = _records()
    got = ta.window_reads(al, chrom_ids, G["vcf_rows"])
    rows = ta.annotate_vcf_with_counts(G["vcf_rows"], got)
    assert "".join("\t".join(r) + "\n" for r in rows) == G["expected_new_table"]


def test_window_reads_against_the_definition_on_random_records():
    """every read with ANY record overlapping [bp-1000, bp+1000) on the locus chromosome (TELR_assembly.py:384-415), checked
    record by record: duplicate loci, loci on unknown chromosomes, windows clipped at 0, windows sharing reads"""
    import numpy as np
    from telr_amd import telr_assembly
    rng = np.random.default_rng(8)
    n = 5000
    al = np.zeros(n, dtype=[("qid", np.int32), ("tid", np.int32), ("ts", np.int32), ("te", np.int32)])
    al["qid"] = rng.integers(0, 700, n); al["tid"] = rng.integers(0, 3, n)
    al["ts"] = rng.integers(0, 60000, n); al["te"] = al["ts"] + rng.integers(1, 9000, n)
    chrom_ids = {"a": 0, "b_x": 1, "c": 2}
    loci = [("a", 300, 400), ("a", 300, 400), ("b_x", 20000, 20001), ("zz", 5, 9), ("c", 59000, 61000), ("a", 1500, 1502), ("b_x", 20500, 20900)]
    loci += [(("a", "b_x", "c")[int(rng.integers(0, 3))], int(s), int(s) + int(rng.integers(0, 50))) for s in rng.integers(0, 62000, 60)]
    got = telr_assembly.window_reads(al, chrom_ids, loci)
    assert len(got) == len(loci)
    for (ch, s, e), g in zip(loci, got):
        bp = telr_assembly.breakpoint(s, e); lo, hi = max(0, bp - 1000), bp + 1000
        want = sorted({int(r["qid"]) for r in al if r["tid"] == chrom_ids.get(ch, -1) and r["ts"] < hi and r["te"] > lo})
        assert list(map(int, g)) == want, (ch, s, e)
    assert telr_assembly.window_reads(al[:0], chrom_ids, loci)[0].size == 0 and telr_assembly.window_reads(al, chrom_ids, []) == []
