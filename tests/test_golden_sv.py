"""Locus-table merge (telr_amd/telr_sv.py) against the reference's merge_vcf post-processing (tests/golden/sv.json)."""
import json
import os

from telr_amd import telr_sv as S

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sv.json")))


def test_bedtools_merge_intermediate():
    assert S.bedtools_merge_rows([list(r) for r in G["table"]], 20) == G["bedtools_merge"]


def test_merge_rows_match_reference():
    got = S.merge_rows([list(r) for r in G["table"]], 20)
    assert len(got) == len(G["merged"])
    for g, e in zip(got, G["merged"]):
        assert g[:8] == e[:8] and g[9:] == e[9:]
        assert sorted(g[8].split(",")) == sorted(e[8].split(","))        # the reference's read order is hash-seed dependent


def test_merge_vcf_files(tmp_path):
    vin, vout = tmp_path / "in.tsv", tmp_path / "out.tsv"
    vin.write_text("".join("\t".join(r) + "\n" for r in G["table"]))
    S.merge_vcf(str(vin), str(vout))
    assert sorted(S.create_loci_set(str(vout))) == G["loci"]
    assert [r[12] for r in S.read_locus_table(str(vout))] == [r[12] for r in G["merged"]]
    fa = tmp_path / "ins.fa"
    S.write_ins_seqs(str(vout), str(fa))
    assert fa.read_text().splitlines()[:2] == [">chr2L_1019_1022", "ACGTAC"]


def test_small_helpers():
    for v, e in G["af_sum"]:
        got = S.af_sum(list(v))
        assert got == e and type(got) is type(e)
    for v, e in G["id_merge"]:
        assert sorted(S.id_merge(v).split(",")) == e
    assert S.get_unique_list(["b", "a", "b"]) == ["b", "a"]
    assert len(S.COLUMNS) == 14
