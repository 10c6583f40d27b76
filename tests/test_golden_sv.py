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


def _write(path, rows):
    path.write_text("".join("\t".join(r) + "\n" for r in rows))


def test_swap_coordinate(tmp_path):
    _write(tmp_path / "raw.tsv", G["parsed"])
    S.swap_coordinate(str(tmp_path / "raw.tsv"), str(tmp_path / "swap.tsv"))
    assert [l.split("\t") for l in (tmp_path / "swap.tsv").read_text().splitlines()] == G["swapped"]


def _same_but_read_order(got_text, want_text, reads_col=8):
    got = [l.split("\t") for l in got_text.splitlines()]
    want = [l.split("\t") for l in want_text.splitlines()]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g[:reads_col] == w[:reads_col] and g[reads_col + 1:] == w[reads_col + 1:]
        assert sorted(g[reads_col].split(",")) == sorted(w[reads_col].split(","))   # set order in the reference


def test_rm_vcf_redundancy_matches_pandas_groupby(tmp_path):
    _write(tmp_path / "swap.tsv", G["swapped"])
    S.rm_vcf_redundancy(str(tmp_path / "swap.tsv"), str(tmp_path / "dedup.tsv"))
    _same_but_read_order((tmp_path / "dedup.tsv").read_text(), G["dedup_text"])


def test_rm_vcf_redundancy_all_capped_af_prints_integer(tmp_path):
    rows = [["c", "1", "2", "5", "1", "0.9", "a", "AC", "r1", "PASS", "0/1", "1", "1"],
            ["c", "1", "2", "5", "1", "0.8", "b", "AC", "r2", "PASS", "0/1", "1", "1"]]
    out = S.dedup_rows(rows)
    assert out == [["c", 1, 2, 5, 2, 1, "a", "AC", "r1,r2", "PASS", "0/1", 2, 2]]


def test_gff_screen_and_proportions(tmp_path):
    gff = tmp_path / "x.out.gff"
    gff.write_text("##gff-version 2\n" + "".join(
        "\t".join([sid, "RepeatMasker", "similarity", str(s), str(e), "12.3", st, ".", 'Target "Motif:%s" 1 %d' % (fam, e - s + 1)]) + "\n"
        for sid, s, e, st, fam in G["rm_gff"]))
    assert S.gff_screen(str(gff)) == G["rm_merged_bed"]


def test_filter_vcf_table_side(tmp_path):
    ins = tmp_path / "dedup.tsv"
    ins.write_text(G["dedup_text"])
    ev = tmp_path / "eval.tsv"
    ev.write_text("")
    out = tmp_path / "o"
    out.mkdir()
    S.filter_vcf(str(ins), str(tmp_path / "filt.tsv"), "lib.fa", str(out), "s+1", 2, str(ev),
                 screen=lambda fa, lib, t: [list(m) for m in G["rm_merged_bed"]])
    assert (tmp_path / "filt.tsv").read_text() == G["filtered_text"]          # incl. the 0.5800000000000001 float sum
    assert sorted(ev.read_text().splitlines()) == G["filter_eval"]
    assert (out / "splus1.vcf_ins.fasta").read_text() == G["ins_fasta"]
    import pytest
    with pytest.raises(ValueError):
        S.filter_vcf(str(ins), str(tmp_path / "f2.tsv"), "lib.fa", str(out), "s", 2, str(ev))
