"""The seven argv shapes of the reference are accepted verbatim (parsing only; no GPU)."""
from telr_amd.cli_mm2 import parse_argv
import pytest


def test_s1_ngmlr():
    o = parse_argv(["ngmlr", "-r", "ref.fa", "-q", "reads.fa", "-x", "ont", "-t", "8", "--rg-id", "S", "--rg-sm", "S", "--rg-lb", "ont", "--no-progress"])
    assert o["preset"] == "ngmlr-ont" and o["sam"] and o["rg"] == ("S", "S", "ont") and (o["target"], o["query"]) == ("ref.fa", "reads.fa")


def test_s2_stage1_minimap2():
    o = parse_argv(["minimap2", "--cs", "--MD", "-Y", "-L", "-ax", "map-pb", "ref.fa", "reads.fa"])
    assert o["preset"] == "map-pb" and o["sam"] and o["cs"] and o["md"] and o["softclip"]


def test_s3_polish():
    o = parse_argv(["minimap2", "-t", "1", "-ax", "map-ont", "-r2k", "cns.fa", "reads.fa"])
    assert o["bw"] == 2000 and o["sam"] and o["threads"] == 1


def test_s4_alt_to_contig():
    o = parse_argv(["minimap2", "-cx", "map-ont", "--secondary=no", "-v", "0", "subj.fa", "qry.fa"])
    assert not o["secondary"] and o["cigar"] and not o["sam"]


def test_s5_library_to_contig():
    o = parse_argv(["minimap2", "-cx", "map-pb", "contig.fa", "lib.fa", "-v", "0", "-t", "4"])
    assert (o["target"], o["query"]) == ("contig.fa", "lib.fa") and o["threads"] == 4


def test_s6_realign():
    o = parse_argv(["minimap2", "-a", "-x", "map-ont", "-v", "0", "contig.fa", "reads.fa"])
    assert o["sam"] and o["preset"] == "map-ont"


def test_s7_flank():
    o = parse_argv(["minimap2", "-cx", "asm10", "-v", "0", "-N", "10", "ref.fa", "flank.fa"])
    assert o["preset"] == "asm10" and o["best_n"] == 10 and o["cigar"]


def test_rejects_unknown():
    with pytest.raises(SystemExit):
        parse_argv(["minimap2", "--splice", "a", "b"])
    with pytest.raises(SystemExit):
        parse_argv(["bwa", "mem"])
