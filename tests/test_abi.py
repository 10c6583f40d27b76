"""The C-ABI shared library: loads without a GPU, exports every symbol include/telr_hip.h declares,
and its presets equal the Python table.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from telr_amd import _lib
from telr_amd._abi import IdxOpt, MapOpt, Aln, Counters
from telr_amd.presets import preset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "telr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(telr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "missing export " + n
    assert set(_lib.EXPORTS) <= set(names)


def test_struct_sizes_match_header():
    assert C.sizeof(Aln) == 88 and C.sizeof(IdxOpt) == 16 and C.sizeof(MapOpt) == 39 * 4 and C.sizeof(Counters) == 96


@pytest.mark.parametrize("name", ["map-ont", "map-pb", "asm10", "ngmlr-ont", "ngmlr-pacbio"])
def test_presets_c_equals_python(name):
    L = _lib.lib()
    io, mo = IdxOpt(), MapOpt()
    assert L.telr_preset(name.encode(), C.byref(io), C.byref(mo)) == 0
    pio, pmo = preset(name)
    for f, _ in IdxOpt._fields_:
        assert getattr(io, f) == getattr(pio, f), f
    for f, _ in MapOpt._fields_:
        a, b = getattr(mo, f), getattr(pmo, f)
        assert a == pytest.approx(b), f
    assert L.telr_preset(b"nope", C.byref(io), C.byref(mo)) < 0


def test_error_strings_and_no_device_is_an_error_not_a_fallback():
    L = _lib.lib()
    assert L.telr_strerror(0) == b"ok"
    assert b"coordinate" in L.telr_strerror(-4)
    import torch
    if not torch.cuda.is_available():
        h = C.c_void_p()
        assert L.telr_init(0, C.byref(h)) < 0       # fails loudly: there is no CPU fallback behind the ABI
        from telr_amd.aligner import Engine
        with pytest.raises(_lib.TelrError):
            Engine(0)
