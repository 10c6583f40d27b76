"""The library's FASTA / FASTQ reader (telr_fasta_load, host code: runs without a GPU) against the plain Python reader on the
bundled files and on awkward inputs: multi-line records, CR LF line ends, empty records, a last line without a line break,
names cut at white space, '@' at the start of a FASTQ quality line."""
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
    assert int(ln.sum()) == len(buf) and (np.diff(off) == ln[:-1]).all()
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
    with pytest.raises(TelrError):
        FastaFile(str(p))
    with pytest.raises(TelrError):
        FastaFile(str(tmp_path / "missing.fa"))
    e = tmp_path / "e.fa"
    e.write_text("")
    f = FastaFile(str(e))
    assert f.n == 0 and f.names == []
