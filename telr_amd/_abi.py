"""ctypes mirrors of the plain-C structs declared in include/telr_hip.h."""
import ctypes as C


class IdxOpt(C.Structure):
    _fields_ = [("k", C.c_int32), ("w", C.c_int32), ("is_hpc", C.c_int32), ("bucket_bits", C.c_int32)]


class MapOpt(C.Structure):
    _fields_ = [
        ("mid_occ_frac", C.c_float), ("min_mid_occ", C.c_int32), ("max_mid_occ", C.c_int32),
        ("max_gap", C.c_int32), ("bw", C.c_int32), ("chain_lookback", C.c_int32), ("min_cnt", C.c_int32),
        ("min_chain_score", C.c_int32), ("chain_gap_q8", C.c_int32), ("chain_skip_q8", C.c_int32),
        ("mask_level", C.c_float), ("pri_ratio", C.c_float), ("best_n", C.c_int32), ("secondary", C.c_int32),
        ("a", C.c_int32), ("b", C.c_int32), ("q", C.c_int32), ("e", C.c_int32), ("q2", C.c_int32), ("e2", C.c_int32),
        ("sc_ambi", C.c_int32), ("zdrop", C.c_int32), ("min_dp_max", C.c_int32), ("min_ksw_len", C.c_int32),
        ("ext_max", C.c_int32), ("ext_band", C.c_int32), ("flags", C.c_int32), ("fill_band_q4", C.c_int32), ("fill_margin", C.c_int32),
        ("vote_len", C.c_int32), ("vote_bin_shift", C.c_int32), ("vote_min", C.c_int32), ("vote_frac_q8", C.c_int32), ("bw_long", C.c_int32),
        ("cx_scale", C.c_int32), ("cx_open", C.c_int32), ("cx_ext_max", C.c_int32), ("cx_ext_min", C.c_int32), ("cx_decay", C.c_int32),
    ]

    def copy(self):
        o = MapOpt()
        C.memmove(C.byref(o), C.byref(self), C.sizeof(MapOpt))
        return o


class Aln(C.Structure):
    _fields_ = [
        ("qid", C.c_int32), ("tid", C.c_int32), ("qlen", C.c_int32), ("qs", C.c_int32), ("qe", C.c_int32),
        ("tlen", C.c_int32), ("ts", C.c_int32), ("te", C.c_int32), ("mlen", C.c_int32), ("blen", C.c_int32),
        ("score", C.c_int32), ("subsc", C.c_int32), ("dp_score", C.c_int32), ("cnt", C.c_int32),
        ("n_sub", C.c_int32), ("parent", C.c_int32), ("n_cigar", C.c_int32), ("flags", C.c_int32),
        ("cigar_off", C.c_int64), ("mapq", C.c_int32), ("n_ambi", C.c_int32),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "query_bases", "minimizers", "probes", "anchors", "chains", "dp_problems", "dp_cells",
        "window_bases", "cigar_ops", "records", "over_queries", "over_ranges")]


# return codes (include/telr_hip.h)
TELR_OK, TELR_E_NODEVICE, TELR_E_HIP, TELR_E_ARG, TELR_E_RANGE, TELR_E_NOMEM, TELR_E_IO = 0, -1, -2, -3, -4, -5, -6
F_PRIMARY, F_SECONDARY, F_SUPPL, F_REV = 1, 2, 4, 8
MF_CIGAR, MF_PER_TARGET, MF_FAITHFUL, MF_KEEP_CIGARS = 1, 2, 4, 8
N_STAGES = 16
N_DPCLS = 25

import numpy as _np

ALN_DTYPE = _np.dtype([(n, _np.int64 if t is C.c_int64 else _np.int32) for n, t in Aln._fields_], align=True)
assert ALN_DTYPE.itemsize == C.sizeof(Aln) == 88, (ALN_DTYPE.itemsize, C.sizeof(Aln))
