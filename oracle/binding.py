"""ctypes binding of oracle/libtelroracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np
from telr_amd._abi import IdxOpt, MapOpt, Aln, Counters, ALN_DTYPE
from telr_amd.fasta import concat

_HERE = os.path.dirname(os.path.abspath(__file__))
# TELR_ORACLE_SO: another build of the same source (tests/test_oracle_asan.py runs the CPU tests against the
# -fsanitize=address,undefined build, `make -C oracle asan`)
_SO = os.environ.get("TELR_ORACLE_SO") or os.path.join(_HERE, "libtelroracle.so")


def build(force=False):
    src = os.path.join(_HERE, "telr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["asan"] if _SO.endswith("_asan.so") else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        L.tor_sketch.restype = i64
        L.tor_sketch.argtypes = [C.c_char_p, i32, C.c_int, C.c_int, C.c_int, C.c_uint32, vp, vp, i64]
        L.tor_index_build.restype = vp
        L.tor_index_build.argtypes = [i32, vp, vp, vp, C.POINTER(IdxOpt)]
        L.tor_index_free.argtypes = [vp]
        L.tor_index_n_mz.restype = i64; L.tor_index_n_mz.argtypes = [vp]
        L.tor_index_n_ent.restype = i64; L.tor_index_n_ent.argtypes = [vp]
        L.tor_index_dump.argtypes = [vp, vp, vp]
        L.tor_mid_occ.restype = i32; L.tor_mid_occ.argtypes = [vp, C.c_float, i32, i32]
        L.tor_map.restype = vp
        L.tor_map.argtypes = [vp, i32, vp, vp, vp, vp, C.POINTER(MapOpt), C.c_int]
        for n in ("tor_result_count", "tor_result_cigar_count", "tor_debug_n_anchor", "tor_debug_n_chain"):
            getattr(L, n).restype = i64; getattr(L, n).argtypes = [vp]
        for n in ("tor_result_alns", "tor_result_cigars", "tor_debug_anchors", "tor_debug_anchor_off",
                  "tor_debug_f", "tor_debug_p", "tor_debug_chains"):
            getattr(L, n).restype = vp; getattr(L, n).argtypes = [vp]
        L.tor_result_counters.argtypes = [vp, C.POINTER(Counters)]
        L.tor_result_free.argtypes = [vp]
        L.tor_nw.restype = i32
        L.tor_nw.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(MapOpt), vp, C.POINTER(i32), i32]
        L.tor_ext.restype = i32
        L.tor_ext.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(MapOpt), vp, C.POINTER(i32), i32,
                              C.POINTER(i32), C.POINTER(i32)]
        L.tor_depth_medians.argtypes = [vp, i64, vp, i32, vp, i32, vp, vp, vp, vp]
        L.tor_consensus.restype = i64; L.tor_consensus.argtypes = [vp, i64, vp, vp, vp, i32, vp, vp, vp, i32, vp, i64, vp, vp]
        L.tor_poa.restype = i64; L.tor_poa.argtypes = [vp, i64, vp, vp, vp, i32, vp, vp, vp, i32, vp, i64, vp, vp]
        _lib = L
    return _lib


def _np_from(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    dt = np.dtype(dtype)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt, count=n).copy()


def sketch(seq, k, w, hpc=0, base=0):
    b = seq.encode() if isinstance(seq, str) else bytes(seq)
    cap = len(b) + 1
    x = np.zeros(cap, np.uint64); y = np.zeros(cap, np.uint32)
    n = lib().tor_sketch(b, len(b), k, w, hpc, base, x.ctypes.data, y.ctypes.data, cap)
    return x[:n].copy(), y[:n].copy()


class OracleIndex:
    def __init__(self, seqs, io):
        self.buf, self.off, self.len = concat(seqs)
        self.io = io
        self.h = lib().tor_index_build(len(seqs), self.buf.ctypes.data, self.off.ctypes.data, self.len.ctypes.data, C.byref(io))
        self.n = len(seqs)

    def dump(self):
        n = lib().tor_index_n_mz(self.h)
        h = np.zeros(n, np.uint64); y = np.zeros(n, np.uint32)
        lib().tor_index_dump(self.h, h.ctypes.data, y.ctypes.data)
        return h, y

    def n_distinct(self):
        return lib().tor_index_n_ent(self.h)

    def mid_occ(self, mo):
        return lib().tor_mid_occ(self.h, mo.mid_occ_frac, mo.min_mid_occ, mo.max_mid_occ)

    def map(self, seqs, mo, qtarget=None, debug=False):
        buf, off, ln = concat(seqs)
        qt = None if qtarget is None else np.ascontiguousarray(qtarget, dtype=np.int32)
        r = lib().tor_map(self.h, len(seqs), buf.ctypes.data, off.ctypes.data, ln.ctypes.data,
                          None if qt is None else qt.ctypes.data, C.byref(mo), 1 if debug else 0)
        L = lib()
        out = {}
        n = L.tor_result_count(r)
        out["alns"] = _np_from(L.tor_result_alns(r), n, ALN_DTYPE)
        out["cigars"] = _np_from(L.tor_result_cigars(r), L.tor_result_cigar_count(r), np.uint32)
        ctr = Counters(); L.tor_result_counters(r, C.byref(ctr))
        out["counters"] = {k: getattr(ctr, k) for k, _ in Counters._fields_}
        if debug:
            na = L.tor_debug_n_anchor(r)
            out["anchors"] = _np_from(L.tor_debug_anchors(r), na, np.uint64)
            out["anchor_off"] = _np_from(L.tor_debug_anchor_off(r), len(seqs) + 1, np.int64)
            out["f"] = _np_from(L.tor_debug_f(r), na, np.int32)
            out["p"] = _np_from(L.tor_debug_p(r), na, np.int32)
            out["chains"] = _np_from(L.tor_debug_chains(r), L.tor_debug_n_chain(r) * 9, np.int32).reshape(-1, 9)
        L.tor_result_free(r)
        return out

    def __del__(self):
        try:
            lib().tor_index_free(self.h)
        except Exception:
            pass


def nw(q, t, mo):
    cap = len(q) + len(t) + 2
    cig = np.zeros(cap, np.uint32); n = C.c_int32(0)
    sc = lib().tor_nw(q.encode(), len(q), t.encode(), len(t), C.byref(mo), cig.ctypes.data, C.byref(n), cap)
    return sc, cig[:n.value].copy()


def ext(q, t, mo):
    cap = len(q) + len(t) + 2
    cig = np.zeros(cap, np.uint32); n = C.c_int32(0); qe = C.c_int32(0); te = C.c_int32(0)
    sc = lib().tor_ext(q.encode(), len(q), t.encode(), len(t), C.byref(mo), cig.ctypes.data, C.byref(n), cap,
                       C.byref(qe), C.byref(te))
    return sc, cig[:n.value].copy(), qe.value, te.value


def depth_medians(alns, cigars, tlens, iv_tid, iv_s, iv_e):
    alns = np.ascontiguousarray(alns); cigars = np.ascontiguousarray(cigars, dtype=np.uint32)
    tl = np.ascontiguousarray(tlens, dtype=np.int32)
    a, b, c = (np.ascontiguousarray(x, dtype=np.int32) for x in (iv_tid, iv_s, iv_e))
    out = np.zeros(len(a), np.float64)
    lib().tor_depth_medians(alns.ctypes.data, len(alns), cigars.ctypes.data, len(tl), tl.ctypes.data, len(a),
                            a.ctypes.data, b.ctypes.data, c.ctypes.data, out.ctypes.data)
    return out


def cigar_str(cigs):
    return "".join("%d%s" % (c >> 4, "MID"[c & 0xf]) for c in cigs)


_NT4 = np.full(256, 4, np.uint8)
for _i, _c in enumerate("ACGT"):
    _NT4[ord(_c)] = _i; _NT4[ord(_c.lower())] = _i
_NT4[ord("U")] = 3; _NT4[ord("u")] = 3


def consensus(alns, cigars, queries, targets, min_depth=3, poa=False):
    """pile-up consensus of the targets from the primary records (spec 3.12), or -- poa=True -- the window partial-order
    consensus (spec 3.13) -> list of str"""
    from telr_amd.fasta import concat
    alns = np.ascontiguousarray(alns); cigars = np.ascontiguousarray(cigars, dtype=np.uint32)
    qb, qo, ql = queries if isinstance(queries, tuple) else concat(queries)
    tb, to, tl = targets if isinstance(targets, tuple) else concat(targets)
    q4 = np.ascontiguousarray(_NT4[np.asarray(qb, np.uint8)]); qo = np.ascontiguousarray(qo, np.int64)
    tb = np.ascontiguousarray(tb, np.uint8); to = np.ascontiguousarray(to, np.int64); tl = np.ascontiguousarray(tl, np.int32)
    cap = int(tl.sum()) * (1 + 8) + 16
    out = np.zeros(cap, np.uint8); ooff = np.zeros(len(tl), np.int64); olen = np.zeros(len(tl), np.int32)
    n = (lib().tor_poa if poa else lib().tor_consensus)(alns.ctypes.data, len(alns), cigars.ctypes.data, q4.ctypes.data, qo.ctypes.data, len(tl), tb.ctypes.data, to.ctypes.data, tl.ctypes.data,
                            int(min_depth), out.ctypes.data, cap, ooff.ctypes.data, olen.ctypes.data)
    assert n <= cap
    return [bytes(out[ooff[i]:ooff[i] + olen[i]]).decode() for i in range(len(tl))]
