"""The library's FASTA / FASTQ reader (telr_fasta_load, host code: runs without a GPU) against the plain Python reader on the
bundled files and on awkward inputs: multi-line records, CR LF line ends, empty records, a last line without a line break,
names cut at white space, '@' at the start of a FASTQ quality line."""
import os

import numpy as np
import pytest

from telr_amd.fasta import read_fasta, FastaFile


def _same(path):
    names, seqs = read_fasta(path)
    f = FastaFile(path)
    assert f.n == len(names)
    assert f.names == names
    assert f.seqs() == seqs
    buf, off, ln = f.triple
    assert int(ln.sum()) == f.bases
    if len(buf) == f.bases:                       # packed end to end (a copy)
        assert (np.diff(off) == ln[:-1]).all()
    else:                                         # the mapped file itself: every sequence inside it, in file order
        assert len(buf) == os.path.getsize(path) and ((off + ln <= len(buf)).all()) and (np.diff(off[ln > 0]) > 0).all()
    f.close()


@pytest.mark.parametrize("name", ["reads.fasta", "ref_38kb.fasta", "library.fasta"])
def test_bundled_files(data_dir, name):
    _same(data_dir + "/" + name)


def test_awkward_fasta(tmp_path):
    p = tmp_path / "a.fa"
    p.write_bytes(b">r1 some description\nACGT\nAC\n\nGT\n>r2\tx\r\nAAAA\r\nCC\r\n>empty\n>r4\nT" )
    f = FastaFile(str(p))
    assert f.names == ["r1", "r2", "empty", "r4"]
    assert f.seqs() == ["ACGTACGT", "AAAACC", "", "T"]
    f.close()
    rng = np.random.default_rng(3)
    # many records of random line widths: the record starts are found by worker threads over slices of the file
    recs = []
    for i in range(5000):
        s = "".join(rng.choice(list("ACGTN"), int(rng.integers(0, 400))))
        w = int(rng.integers(1, 90))
        recs.append((">q%d/%d extra" % (i, w), s, w))
    q = tmp_path / "b.fa"
    with open(q, "w") as fh:
        for h, s, w in recs:
            fh.write(h + "\n")
            for k in range(0, len(s), w):
                fh.write(s[k:k + w] + "\n")
    _same(str(q))


def test_single_line_records_are_used_in_place(tmp_path, monkeypatch):
    """every sequence on one line: the base buffer is the mapped file (nothing copied); one folded or CR LF record: a packed copy;
    both give the same names and sequences as the Python reader, and the same as the copying path (TELR_AB=fasta_copy)"""
    rng = np.random.default_rng(5)
    seqs = ["".join(rng.choice(list("ACGTN"), int(rng.integers(0, 300)))) for _ in range(3000)]
    p = tmp_path / "one.fa"
    with open(p, "w") as fh:
        for i, s in enumerate(seqs):
            fh.write(">s%d desc\n%s\n" % (i, s))
            if i % 97 == 0:
                fh.write("\n")                   # a blank line between records
    f = FastaFile(str(p))
    buf, off, ln = f.triple
    assert len(buf) == os.path.getsize(p) and f.bases == sum(len(x) for x in seqs)
    assert f.seqs() == seqs and f.names == ["s%d" % i for i in range(len(seqs))]
    f.close()
    _same(str(p))
    q = tmp_path / "folded.fa"
    q.write_text(p.read_text() + ">last\nACGT\nAC\n")
    g = FastaFile(str(q))
    assert len(g.triple[0]) == g.bases == sum(len(x) for x in seqs) + 6 and g.seqs() == seqs + ["ACGTAC"]
    g.close()
    # last record without a line break, blank line before a sequence, empty record
    r = tmp_path / "edge.fa"
    r.write_bytes(b">a\n\nACGT\n>b\n>c\nGG")
    h = FastaFile(str(r))
    assert h.seqs() == ["ACGT", "", "GG"] and len(h.triple[0]) == os.path.getsize(r)
    h.close()
    _same(str(r))


def test_fastq(tmp_path):
    p = tmp_path / "a.fq"
    p.write_text("@r1 d\nACGT\n+\n@III\n@r2\nGG\n+r2\n@@\n@r3\n\n+\n\n@r4\nTTTT\n+\n!!!!")
    f = FastaFile(str(p))
    assert f.names == ["r1", "r2", "r3", "r4"] and f.seqs() == ["ACGT", "GG", "", "TTTT"]
    f.close()
    _same(str(p))


def test_errors(tmp_path):
    from telr_amd._lib import TelrError
    p = tmp_path / "x.txt"
    p.write_text("hello\n")
    from telr_amd._abi import TELR_E_ARG, TELR_E_IO
    from telr_amd.fasta import load
    with pytest.raises(TelrError) as ei:
        FastaFile(str(p))
    assert ei.value.code == TELR_E_ARG                  # a layout the parser refuses: fasta.load() hands the file to the Python reader
    assert load(str(p)) is None
    with pytest.raises(TelrError) as ei:
        FastaFile(str(tmp_path / "missing.fa"))
    assert ei.value.code == TELR_E_IO                   # an I/O failure has its own code (round 6) ...
    with pytest.raises(TelrError):
        load(str(tmp_path / "missing.fa"))              # ... and is NOT retried on the slow path
    e = tmp_path / "e.fa"
    e.write_text("")
    f = FastaFile(str(e))
    assert f.n == 0 and f.names == []


def test_copying_path_behind_its_switch(tmp_path):
    """TELR_AB=fasta_copy (read once per process): single-line records are copied into a packed buffer as folded ones are; same names and sequences"""
    import subprocess, sys
    p = tmp_path / "one.fa"
    p.write_text("".join(">s%d\n%s\n" % (i, "ACGT" * (i % 7 + 1)) for i in range(200)))
    code = "import sys; sys.path.insert(0, %r)\nfrom telr_amd.fasta import FastaFile\nf = FastaFile(sys.argv[1])\nprint(len(f.triple[0]) == f.bases, f.seqs() == ['ACGT' * (i %% 7 + 1) for i in range(200)])" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env, want in (({}, b"False True"), ({"TELR_AB": "fasta_copy"}, b"True True")):
        r = subprocess.run([sys.executable, "-c", code, str(p)], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0 and r.stdout.strip() == want, (env, r.stdout, r.stderr.decode()[-1000:])


def test_leading_and_trailing_blank_lines_and_multi_line_fastq(tmp_path):
    """files minimap2 / ngmlr accept: white space before the first record, blank lines after the last FASTQ
    record -- read by the C parser; a multi-line FASTQ -- refused by it and handed to the Python reader by fasta.load()"""
    from telr_amd.fasta import load
    p = tmp_path / "lead.fa"
    p.write_bytes(b"\n \n>r1\nACGT\n>r2\nGG\n\n")
    f = FastaFile(str(p))
    assert f.names == ["r1", "r2"] and f.seqs() == ["ACGT", "GG"]
    f.close()
    q = tmp_path / "tail.fq"
    q.write_bytes(b"\n@a x\nACGT\n+\n@III\n@b\nTT\n+\nII\n\n\n")
    f = FastaFile(str(q))
    assert f.names == ["a", "b"] and f.seqs() == ["ACGT", "TT"]
    f.close()
    e = tmp_path / "blank.fa"
    e.write_bytes(b"\n\n  \n")
    f = FastaFile(str(e))
    assert f.n == 0
    f.close()
    m = tmp_path / "multi.fq"
    m.write_bytes(b"@a\nACGT\nACGT\n+\nIIII\nIIII\n")
    assert load(str(m)) is None                    # the caller's Python reader takes over
    assert load(str(q)) is not None
