"""CPU checks of the host half of the device deflate (telr_amd/csrc/bam_dev.hip.h): length-limited Huffman code lengths are
complete prefix codes, and a dynamic-block stream put together from them -- header bit string, literals, distance-1 matches
cut at 64-byte pieces, exactly what k_bgzf_deflate emits for a one-segment block -- is inflated by zlib to the input."""
import ctypes as C
import zlib

import numpy as np
import pytest

from telr_amd import _lib


def _lens(freq, maxlen):
    L = _lib.lib()
    f = np.ascontiguousarray(freq, np.uint32)
    out = np.zeros(len(f), np.uint8)
    assert L.telr_debug_huff(f.ctypes.data, len(f), maxlen, out.ctypes.data) == 0
    return out


@pytest.mark.parametrize("seed", range(6))
def test_length_limited_codes_are_complete(seed):
    rng = np.random.default_rng(seed)
    for n, maxlen in ((286, 15), (19, 7), (30, 15), (286, 9)):
        kind = seed % 3
        if kind == 0:
            f = rng.integers(1, 1000, n)
        elif kind == 1:
            f = (2.0 ** rng.uniform(0, 28, n)).astype(np.int64)          # Fibonacci-like skew: unlimited depths far beyond maxlen
        else:
            f = rng.integers(0, 3, n) * rng.integers(1, 10 ** 6, n)          # many zeros
        f = np.minimum(f, 2 ** 31 - 1)
        ln = _lens(f, maxlen)
        used = f > 0
        assert (ln[~used] == 0).all() and (ln[used] >= 1).all() and ln.max() <= maxlen
        if used.sum() >= 2:
            assert sum(2.0 ** -int(x) for x in ln[used]) == 1.0          # complete
            # a rarer symbol never has a shorter code
            order = np.argsort(f[used], kind="stable")
            assert (np.diff(ln[used][order].astype(int)) <= 0).all()


def test_optimal_when_unconstrained():
    f = np.array([45, 13, 12, 16, 9, 5])          # the textbook example: lengths 1,3,3,3,4,4
    assert sorted(_lens(f, 15).tolist()) == [1, 3, 3, 3, 4, 4]


def _host_deflate(data):
    L = _lib.lib()
    src = np.frombuffer(data, np.uint8)
    out = np.zeros(len(src) * 2 + 1024, np.uint8)
    n = C.c_int32(0)
    assert L.telr_debug_deflate_host(src.ctypes.data if len(src) else out.ctypes.data, len(src), out.ctypes.data, len(out), C.byref(n)) == 0
    return out[:n.value].tobytes()


@pytest.mark.parametrize("seed", range(5))
def test_host_stream_inflates_to_input(seed):
    rng = np.random.default_rng(100 + seed)
    parts = [rng.integers(0, 256, 3000, dtype=np.uint8).tobytes(),                       # every byte value
             b"\xff" * 9000,                                                              # a QUAL run across many pieces
             bytes(rng.choice([0x11, 0x12, 0x14, 0x18, 0x21, 0x88], 5000).astype(np.uint8)),   # SEQ-like
             b"".join(b"%d%s" % (rng.integers(0, 60), rng.choice([b"A", b"^CG", b":", b"*ag"])) for _ in range(2000)),
             np.repeat(rng.integers(0, 256, 400, dtype=np.uint8), rng.integers(1, 9, 400)).tobytes(),   # short runs: 1..8
             b"ab" * 100, b"\x00" * 2, b"\x00" * 3, b"\x01", b""]
    order = rng.permutation(len(parts))
    data = b"".join(parts[i] for i in order)
    z = _host_deflate(data)
    assert zlib.decompress(z, -15) == data
    assert len(z) < len(data)
    assert zlib.decompress(_host_deflate(b""), -15) == b""
    assert zlib.decompress(_host_deflate(b"\xff" * 70000), -15) == b"\xff" * 70000
