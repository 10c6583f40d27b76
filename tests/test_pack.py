"""Host-side 2-bit packing of sequences (telr_seqset_create): the AVX2 packer and the 64-bit word packer against a plain
Python statement of the layout (base i of a sequence: code bits 2i..2i+1 of the code words, bit i of the ambiguity words;
A 0, C 1, G 2, T/U 3 in either case, anything else ambiguous with code 0; padding up to the 64-base group is A)."""
import ctypes as C

import numpy as np
import pytest

from telr_amd import _lib

CODE = {c: v for v, cs in enumerate(("Aa", "Cc", "Gg", "TtUu")) for c in cs.encode()}


def _expect(b):
    n = len(b)
    g = (n + 63) // 64
    code = np.zeros(g * 4, np.uint32)
    amb = np.zeros(g * 2, np.uint32)
    for i, c in enumerate(b):
        v = CODE.get(c)
        if v is None:
            amb[i >> 5] |= np.uint32(1 << (i & 31))
        else:
            code[i >> 4] |= np.uint32(v << (2 * (i & 15)))
    return code, amb


def _pack(L, b, mode):
    g = (len(b) + 63) // 64
    code = np.full(g * 4 + 1, 0xdeadbeef, np.uint32)
    amb = np.full(g * 2 + 1, 0xdeadbeef, np.uint32)
    rc = L.telr_debug_pack(bytes(b), len(b), mode, code.ctypes.data, amb.ctypes.data)
    assert code[-1] == 0xdeadbeef and amb[-1] == 0xdeadbeef          # nothing written past the last group
    return rc, code[:-1], amb[:-1]


@pytest.mark.parametrize("mode", [0, 1])
def test_packers_match_the_layout(mode):
    L = _lib.lib()
    rng = np.random.default_rng(5)
    cases = [b"", b"A", b"acgtuACGTUnN-*", bytes(range(256)), bytes(range(255, -1, -1)) * 3]
    for n in (1, 7, 8, 9, 31, 32, 33, 63, 64, 65, 127, 128, 129, 1000):
        cases.append(bytes(rng.choice(np.frombuffer(b"ACGTacgtNnUuRYKM", np.uint8), n)))
        cases.append(bytes(rng.integers(0, 256, n, dtype=np.uint8)))
    for b in cases:
        rc, code, amb = _pack(L, b, mode)
        if rc != 0:
            pytest.skip("host without AVX2")
        ec, ea = _expect(b)
        np.testing.assert_array_equal(code, ec, err_msg=repr(b[:20]))
        np.testing.assert_array_equal(amb, ea, err_msg=repr(b[:20]))
