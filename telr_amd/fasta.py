"""Minimal FASTA/FASTQ reader (stands in for Bio.SeqIO at the call sites of the path)."""
import gzip
import numpy as np


def _open(path):
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "r")


def read_fasta(path):
    """-> (names, seqs) ; FASTA or FASTQ, header truncated at first whitespace."""
    names, seqs = [], []
    with _open(path) as fh:
        first = fh.read(1)
        if not first:
            return names, seqs
        fh.seek(0)
        if first == "@":
            while True:
                h = fh.readline()
                if not h:
                    break
                s = fh.readline().strip()
                fh.readline()
                fh.readline()
                names.append(h[1:].split()[0])
                seqs.append(s)
        else:
            cur = []
            for line in fh:
                if line.startswith(">"):
                    if names:
                        seqs.append("".join(cur))
                    names.append(line[1:].split()[0] if len(line) > 1 and line[1:].split() else "")
                    cur = []
                else:
                    cur.append(line.strip())
            if names:
                seqs.append("".join(cur))
    return names, seqs


def concat(seqs):
    """list of str/bytes -> (uint8 buffer, int64 offsets, int32 lengths) as the C ABI wants them."""
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    lens = np.array([len(b) for b in bs], dtype=np.int32)
    off = np.zeros(len(bs), dtype=np.int64)
    if len(bs) > 1:
        off[1:] = np.cumsum(lens[:-1], dtype=np.int64)
    buf = np.frombuffer(b"".join(bs), dtype=np.uint8) if bs else np.zeros(0, np.uint8)
    return buf, off, lens


_COMP = bytes.maketrans(b"ACGTUNacgtun", b"TGCAANtgcaan")


def revcomp(s):
    if isinstance(s, str):
        return s.encode().translate(_COMP)[::-1].decode()
    return bytes(s).translate(_COMP)[::-1]
