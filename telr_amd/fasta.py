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


class FastaFile:
    """A FASTA / FASTQ file parsed by the library (telr_fasta_load: worker threads over the mapped file) into the arrays the C
    ABI takes.  `.triple` = (base buffer, offsets, lengths) as numpy views of library memory (valid while this object lives),
    `.names_c` = the C array of names for the writers, `.names` = the same as Python strings (built on first use)."""

    def __init__(self, path):
        import ctypes as C
        from . import _lib
        self.L = _lib.lib()
        h = C.c_void_p()
        rc = self.L.telr_fasta_load(str(path).encode(), C.byref(h))
        if rc != 0:
            raise _lib.TelrError("telr_fasta_load(%s): %s" % (path, self.L.telr_strerror(rc).decode()), code=rc)
        self.h = h
        self.n = int(self.L.telr_fasta_count(h))
        nb = int(self.L.telr_fasta_extent(h))        # the packed bases, or the whole mapped file when its sequences are used in place
        self.bases = int(self.L.telr_fasta_bases(h))

        def view(ptr, count, dt):
            if not count or not ptr:
                return np.zeros(0, dt)
            return np.frombuffer((C.c_char * (count * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt, count=count)
        self.triple = (view(self.L.telr_fasta_seq(h), nb, np.uint8), view(self.L.telr_fasta_off(h), self.n, np.int64), view(self.L.telr_fasta_len(h), self.n, np.int32))
        self.names_c = C.cast(self.L.telr_fasta_names(h), C.POINTER(C.c_char_p * max(1, self.n))).contents if self.n else (C.c_char_p * 1)()
        self._names = None

    @property
    def names(self):
        if self._names is None:
            self._names = [self.names_c[i].decode() for i in range(self.n)]
        return self._names

    def seqs(self):
        buf, off, ln = self.triple
        return [bytes(buf[off[i]:off[i] + ln[i]]).decode() for i in range(self.n)]

    def close(self):
        if getattr(self, "h", None):
            self.triple = None; self.names_c = None
            self.L.telr_fasta_free(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load(path):
    """FastaFile for plain files; gzip goes through the Python reader (names, seqs lists wrapped the same way)"""
    if str(path).endswith(".gz"):
        return None
    from . import _lib
    try:
        return FastaFile(path)
    except _lib.TelrError as e:
        # a layout the C parser refuses (multi-line FASTQ, a record that does not start with '>' / '@': TELR_E_ARG): the Python
        # reader decides.  Anything else -- a record too long for the engine (TELR_E_RANGE), an I/O failure (TELR_E_IO: open / stat /
        # mmap), no memory -- is the caller's to see: the slow path would only fail later with an unrelated message.
        from ._abi import TELR_E_ARG
        if getattr(e, "code", None) != TELR_E_ARG:
            raise
        return None
