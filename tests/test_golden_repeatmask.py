"""The RepeatMasker hand-off glue (telr_amd/telr_te.py: parse_rm_out, gff3tobed) against outputs of the reference's own
functions (tests/golden/repeatmask.json, captured by tools/capture_goldens.py from src/telr/TELR_te.py:436-494), and the
checksum-keyed cache of the whole-reference masking (telr.py:132-144 runs RepeatMasker on the full reference on every call)."""
import json
import os

import pytest

from telr_amd import telr_te

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "repeatmask.json")


def test_parse_rm_out_and_gff3tobed_equal_the_reference(tmp_path):
    g = json.load(open(GOLD))
    gff, gff3, bed = (str(tmp_path / n) for n in ("x.out.gff", "x.out.gff3", "x.te.bed"))
    open(gff, "w").write(g["rm_out_gff"])
    telr_te.parse_rm_out(gff, gff3)
    assert open(gff3).read() == g["gff3"]
    telr_te.gff3tobed(gff3, bed)
    assert open(bed).read() == g["bed"]


def _fake_repeatmasker(calls, hits=True):
    g = json.load(open(GOLD))

    def run(argv):
        calls.append(list(argv))
        assert argv[0] == "RepeatMasker" and argv[1] == "-dir" and argv[3:9] == telr_te.RM_FLAGS and argv[9] == "-lib" and argv[11] == "-pa"
        outdir, ref = argv[2], argv[-1]
        base = os.path.join(outdir, os.path.basename(ref))
        if hits:
            open(base + ".masked", "w").write(">chr\nNNNN\n")
            open(base + ".out.gff", "w").write(g["rm_out_gff"])
        else:
            open(base + ".out", "w").write("There were no repetitive sequences detected in " + ref + "\n")
        return 0
    return run


def test_repeatmask_cache_runs_the_tool_once_per_reference_and_library(tmp_path):
    g = json.load(open(GOLD))
    ref, lib, ref2 = (str(tmp_path / n) for n in ("ref.fa", "lib.fa", "ref2.fa"))
    open(ref, "w").write(">chr\nACGTACGT\n"); open(lib, "w").write(">te\nACGT\n"); open(ref2, "w").write(">chr\nACGTACGA\n")
    cache = str(tmp_path / "cache")
    calls = []
    run = _fake_repeatmasker(calls)
    m1, g1 = telr_te.repeatmask(ref, lib, str(tmp_path / "o1"), 4, cache_dir=cache, runner=run)
    assert len(calls) == 1 and calls[0][-3:] == ["-pa", "4", ref] and open(g1).read() == g["gff3"] and m1.endswith("ref.fa.masked")
    m2, g2 = telr_te.repeatmask(ref, lib, str(tmp_path / "o2"), 8, cache_dir=cache, runner=run)         # same bytes: no second run
    assert len(calls) == 1 and open(g2).read() == g["gff3"] and open(m2).read() == open(m1).read() and m2.startswith(str(tmp_path / "o2"))
    telr_te.repeatmask(ref2, lib, str(tmp_path / "o3"), 4, cache_dir=cache, runner=run)                # another reference: runs
    assert len(calls) == 2
    telr_te.repeatmask(ref, lib, str(tmp_path / "o4"), 4, runner=run)                                  # no cache: runs, as the reference does
    assert len(calls) == 3
    assert telr_te.repeatmask_key(ref, lib) != telr_te.repeatmask_key(ref2, lib) != telr_te.repeatmask_key(lib, ref)


def test_repeatmask_without_hits_and_failure(tmp_path):
    ref, lib = str(tmp_path / "ref.fa"), str(tmp_path / "lib.fa")
    open(ref, "w").write(">chr\nACGT\n"); open(lib, "w").write(">te\nTTTT\n")
    cache = str(tmp_path / "cache")
    calls = []
    m, g3 = telr_te.repeatmask(ref, lib, str(tmp_path / "o"), 1, cache_dir=cache, runner=_fake_repeatmasker(calls, hits=False))
    assert (m, g3) == (ref, None) and len(calls) == 1                  # TELR_te.py:415-420
    m, g3 = telr_te.repeatmask(ref, lib, str(tmp_path / "p"), 1, cache_dir=cache, runner=_fake_repeatmasker(calls, hits=False))
    assert (m, g3) == (ref, None) and len(calls) == 1                  # the empty answer is cached as well
    with pytest.raises(SystemExit):                                    # the tool wrote nothing: the reference exits (:428-431)
        telr_te.repeatmask(ref, str(tmp_path / "lib.fa"), str(tmp_path / "q"), 1, runner=lambda argv: 1)
