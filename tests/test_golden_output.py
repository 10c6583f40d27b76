"""Output writers (telr_amd/telr_output.py) against files the reference's TELR_output.py wrote for the same inputs
(tests/golden/output.json, made by tools/capture_goldens.py; the ##fileDate stamp and the temp path of the reference
FASTA are masked there as DATE / REF.fa)."""
import json
import os

import pytest

from telr_amd import telr_output as O

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "output.json")))


@pytest.fixture()
def stage(tmp_path):
    d = tmp_path
    (d / "contigs.fa").write_text(G["contigs_fa"])
    (d / "te.bed").write_text(G["annotation_bed"])
    (d / "te.fa").write_text(G["te_fa"])
    (d / "vcf.tsv").write_text(G["vcf_parsed"])
    (d / "REF.fa").write_text("")
    (d / "REF.fa.fai").write_text(G["ref_fai"])
    return d


def run(stage, lift, sample):
    out = stage / ("out_" + sample)
    out.mkdir(exist_ok=True)
    cwd = os.getcwd()
    os.chdir(stage)                     # so that ##reference= prints the masked relative name
    try:
        O.generate_output(lift, G["te_freq"], "te.fa", "vcf.tsv", "te.bed", "contigs.fa", str(out), sample, "REF.fa", today="DATE")
    finally:
        os.chdir(cwd)
    return {p.name: p.read_text() for p in out.iterdir()}


def test_all_files_byte_identical(stage):
    got = run(stage, G["liftover"], "s")
    for name in ("s.telr.json", "s.telr.expanded.json", "s.telr.te.fasta", "s.telr.contig.fasta", "s.telr.vcf", "s.telr.bed"):
        assert got[name] == G["files"][name], name


def test_report_path_and_list_agree(stage):
    p = stage / "lift.json"
    p.write_text(json.dumps(G["liftover"]))
    assert run(stage, str(p), "s") == run(stage, G["liftover"], "s")


def test_no_non_reference_rows(stage):
    got = run(stage, [G["liftover"][2]], "e")
    for name in ("e.telr.vcf", "e.telr.bed", "e.telr.json"):
        assert got[name] == G["files"][name], name
    assert got["e.telr.te.fasta"] == "" and got["e.telr.contig.fasta"] == ""


def test_all_missing_tsd_prints_none(stage):
    got = run(stage, [G["liftover"][1]], "n")
    assert got["n.telr.vcf"] == G["files"]["n.telr.vcf"]
    assert "TSD_LEN=None;TSD_SEQ=None" in got["n.telr.vcf"]


def test_swapped_dr_dv_kept(stage):
    body = [l for l in run(stage, G["liftover"], "s")["s.telr.vcf"].splitlines() if not l.startswith("#")]
    f = body[0].split("\t")
    assert f[8] == "GT:DR:DV" and f[9] == "0/1:12:5"        # variant reads (12) sit in the DR slot, as in the reference


def test_column_text_rules():
    assert O._column_text([1, 2]) == ["1", "2"]
    assert O._column_text([1, None]) == ["1.0", "nan"]
    assert O._column_text([0.75, 1]) == ["0.75", "1.0"]
    assert O._column_text(["a", None]) == ["a", "None"]
    assert O._column_text([None, None]) == ["None", "None"]


def test_fai_matches_reference_index(tmp_path):
    """write_fai reproduces the .fai the capture script wrote for its reference FASTA (names and lengths feed ##contig)."""
    fa = tmp_path / "r.fa"
    fa.write_text(">chrA desc\nACGTACGTAC\nACGTACGTAC\nACG\n>chrB\nAC\n")
    O.write_fai(str(fa))
    assert (tmp_path / "r.fa.fai").read_text() == "chrA\t23\t11\t10\t11\nchrB\t2\t43\t2\t3\n"
    assert O.get_contig_info(str(fa)) == ["##contig=<ID=chrA,length=23>", "##contig=<ID=chrB,length=2>"]
    bad = tmp_path / "bad.fa"
    bad.write_text(">x\nACG\nACGT\n")
    with pytest.raises(ValueError):
        O.write_fai(str(bad))
