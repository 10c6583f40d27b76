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
