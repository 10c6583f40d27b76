"""ctypes binding of libtelrhip.so (include/telr_hip.h).  There is no CPU fallback:
if the HIP library is missing or no device is usable, calls raise."""
import ctypes as C
import os
from ._abi import IdxOpt, MapOpt, Aln, Counters, N_STAGES

# up to eight engine streams are busy at a time; with RCCL in the same process the HIP default of 4 hardware queues
# serialises them (bench.py: 35.8 vs 29.4 ms per step).  Only effective if no HIP call has been made yet.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("TELR_LIB") or os.path.join(_HERE, "libtelrhip.so")          # TELR_LIB: another build of the same library (tools/ab_build.sh)

EXPORTS = [
    "telr_init", "telr_destroy", "telr_strerror", "telr_last_error", "telr_device_name", "telr_preset",
    "telr_seqset_create", "telr_seqset_subset", "telr_seqset_subset_rc", "telr_seqset_free", "telr_seqset_bases", "telr_seqset_count",
    "telr_index_build", "telr_index_free", "telr_index_stats", "telr_map",
    "telr_result_from_arrays", "telr_result_count", "telr_result_alns", "telr_result_cigar_count", "telr_result_cigars", "telr_result_wait", "telr_result_free",
    "telr_init_background", "telr_release_scratch", "telr_device_mem", "telr_write_paf", "telr_write_sam", "telr_write_bam", "telr_write_bam_dev", "telr_bam_prepare", "telr_bam_release_wait", "telr_bam_discard", "telr_write_bam_slice", "telr_bam_segment_info", "telr_bam_segment_entries", "telr_bam_segment_write", "telr_bam_segment_free", "telr_bai_write", "telr_result_from_device_cigars", "telr_seqset_packed", "telr_seqset_from_packed", "telr_consensus_build", "telr_poa_build", "telr_consensus_count", "telr_consensus_seq", "telr_consensus_off", "telr_consensus_len", "telr_consensus_free", "telr_fasta_load", "telr_fasta_count", "telr_fasta_bases", "telr_fasta_extent", "telr_fasta_seq", "telr_fasta_off", "telr_fasta_len", "telr_fasta_names", "telr_fasta_free", "telr_depth_medians", "telr_window_reads", "telr_stage_ms", "telr_stage_name", "telr_last_counters", "telr_last_dp_classes",
]

_lib = None


class TelrError(RuntimeError):
    """`code` = the library's TELR_E_* return code when the error came from a C-ABI call (None otherwise)"""
    def __init__(self, msg, code=None):
        RuntimeError.__init__(self, msg)
        self.code = code


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise TelrError("libtelrhip.so is not built (run `python -m telr_amd.build`); there is no CPU fallback")
    L = C.CDLL(SO_PATH)
    vp, i32, i64, cp = C.c_void_p, C.c_int32, C.c_int64, C.c_char_p
    L.telr_init.restype = C.c_int; L.telr_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.telr_init_background.restype = C.c_int; L.telr_init_background.argtypes = [C.c_int, C.POINTER(vp)]
    L.telr_release_scratch.restype = C.c_int; L.telr_release_scratch.argtypes = [vp]
    L.telr_device_mem.restype = C.c_int; L.telr_device_mem.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.telr_destroy.restype = None; L.telr_destroy.argtypes = [vp]
    L.telr_strerror.restype = cp; L.telr_strerror.argtypes = [C.c_int]
    L.telr_last_error.restype = cp; L.telr_last_error.argtypes = [vp]
    L.telr_device_name.restype = C.c_int; L.telr_device_name.argtypes = [vp, cp, C.c_int]
    L.telr_preset.restype = C.c_int; L.telr_preset.argtypes = [cp, C.POINTER(IdxOpt), C.POINTER(MapOpt)]
    L.telr_seqset_create.restype = C.c_int; L.telr_seqset_create.argtypes = [vp, i32, vp, vp, vp, C.POINTER(vp)]
    L.telr_seqset_subset.restype = C.c_int; L.telr_seqset_subset.argtypes = [vp, vp, i32, vp, C.POINTER(vp)]
    L.telr_seqset_subset_rc.restype = C.c_int; L.telr_seqset_subset_rc.argtypes = [vp, vp, i32, vp, vp, C.POINTER(vp)]
    L.telr_seqset_packed.restype = C.c_int; L.telr_seqset_packed.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), C.POINTER(i64)]
    L.telr_seqset_from_packed.restype = C.c_int; L.telr_seqset_from_packed.argtypes = [vp, i32, vp, vp, i64, vp, i64, C.POINTER(vp)]
    L.telr_seqset_free.restype = None; L.telr_seqset_free.argtypes = [vp]
    L.telr_seqset_bases.restype = i64; L.telr_seqset_bases.argtypes = [vp]
    L.telr_seqset_count.restype = i32; L.telr_seqset_count.argtypes = [vp]
    L.telr_index_build.restype = C.c_int; L.telr_index_build.argtypes = [vp, vp, C.POINTER(IdxOpt), C.POINTER(vp)]
    L.telr_index_free.restype = None; L.telr_index_free.argtypes = [vp]
    L.telr_index_stats.restype = C.c_int; L.telr_index_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.telr_map.restype = C.c_int; L.telr_map.argtypes = [vp, vp, vp, vp, C.POINTER(MapOpt), C.POINTER(vp)]
    L.telr_result_from_arrays.restype = C.c_int; L.telr_result_from_arrays.argtypes = [vp, vp, i64, vp, i64, C.POINTER(vp)]
    L.telr_result_count.restype = i64; L.telr_result_count.argtypes = [vp]
    L.telr_result_alns.restype = vp; L.telr_result_alns.argtypes = [vp]
    L.telr_result_cigar_count.restype = i64; L.telr_result_cigar_count.argtypes = [vp]
    L.telr_result_cigars.restype = vp; L.telr_result_cigars.argtypes = [vp]
    L.telr_result_wait.restype = C.c_int; L.telr_result_wait.argtypes = [vp]
    L.telr_result_free.restype = None; L.telr_result_free.argtypes = [vp]
    L.telr_depth_medians.restype = C.c_int
    L.telr_depth_medians.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, vp]
    L.telr_window_reads.restype = C.c_int; L.telr_window_reads.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp, i64, vp]
    L.telr_write_paf.restype = C.c_int; L.telr_write_paf.argtypes = [vp, vp, vp, C.c_int, cp, C.c_int]
    L.telr_write_sam.restype = C.c_int
    L.telr_write_sam.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, cp, cp, cp, cp, cp]
    L.telr_write_bam.restype = C.c_int
    L.telr_write_bam.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, cp, cp, cp, cp, cp, i32, i32]
    L.telr_write_bam_dev.restype = C.c_int
    L.telr_write_bam_dev.argtypes = [vp, vp, vp, vp, vp, vp, i32, cp, cp, cp, cp, cp, i32, i32]
    L.telr_fasta_load.restype = C.c_int; L.telr_fasta_load.argtypes = [cp, C.POINTER(vp)]
    L.telr_fasta_count.restype = i32; L.telr_fasta_count.argtypes = [vp]
    L.telr_fasta_bases.restype = i64; L.telr_fasta_bases.argtypes = [vp]
    L.telr_fasta_extent.restype = i64; L.telr_fasta_extent.argtypes = [vp]
    for fn in ("telr_fasta_seq", "telr_fasta_off", "telr_fasta_len", "telr_fasta_names"):
        getattr(L, fn).restype = vp; getattr(L, fn).argtypes = [vp]
    L.telr_fasta_free.restype = None; L.telr_fasta_free.argtypes = [vp]
    L.telr_consensus_build.restype = C.c_int; L.telr_consensus_build.argtypes = [vp, vp, vp, vp, i32, C.POINTER(vp)]
    L.telr_poa_build.restype = C.c_int; L.telr_poa_build.argtypes = [vp, vp, vp, vp, i32, C.POINTER(vp)]
    L.telr_consensus_count.restype = i32; L.telr_consensus_count.argtypes = [vp]
    for fn in ("telr_consensus_seq", "telr_consensus_off", "telr_consensus_len"):
        getattr(L, fn).restype = vp; getattr(L, fn).argtypes = [vp]
    L.telr_consensus_free.restype = None; L.telr_consensus_free.argtypes = [vp]
    L.telr_bam_prepare.restype = C.c_int; L.telr_bam_prepare.argtypes = [vp, cp, i64]
    L.telr_bam_release_wait.restype = C.c_int; L.telr_bam_release_wait.argtypes = []
    L.telr_bam_discard.restype = C.c_int; L.telr_bam_discard.argtypes = [vp]
    L.telr_write_bam_slice.restype = C.c_int; L.telr_write_bam_slice.argtypes = [vp, vp, vp, vp, vp, vp, i32, cp, cp, cp, cp, vp, i32, i32, C.POINTER(vp)]
    L.telr_bam_segment_info.restype = C.c_int; L.telr_bam_segment_info.argtypes = [vp, vp]
    L.telr_bam_segment_entries.restype = C.c_int; L.telr_bam_segment_entries.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    L.telr_bam_segment_write.restype = C.c_int; L.telr_bam_segment_write.argtypes = [vp, vp, cp, i64, i32]
    L.telr_bam_segment_free.restype = None; L.telr_bam_segment_free.argtypes = [vp]
    L.telr_bai_write.restype = C.c_int; L.telr_bai_write.argtypes = [cp, i64, vp, vp, vp, vp, C.c_uint64, i64, i32, vp]
    L.telr_result_from_device_cigars.restype = C.c_int; L.telr_result_from_device_cigars.argtypes = [vp, vp, i64, vp, i64, C.POINTER(vp)]
    L.telr_debug_bam_twin.restype = C.c_int; L.telr_debug_bam_twin.argtypes = []
    L.telr_debug_result_twin.restype = i64; L.telr_debug_result_twin.argtypes = [vp, vp, i64]
    L.telr_debug_bam_sink_ms.restype = C.c_int; L.telr_debug_bam_sink_ms.argtypes = [vp]
    L.telr_debug_bam_ms.restype = C.c_int; L.telr_debug_bam_ms.argtypes = [vp]
    L.telr_debug_huff.restype = C.c_int; L.telr_debug_huff.argtypes = [vp, i32, i32, vp]
    L.telr_debug_deflate_host.restype = C.c_int; L.telr_debug_deflate_host.argtypes = [vp, i32, vp, i32, vp]
    L.telr_stage_ms.restype = C.c_int; L.telr_stage_ms.argtypes = [vp, vp]
    L.telr_stage_name.restype = cp; L.telr_stage_name.argtypes = [C.c_int]
    L.telr_last_counters.restype = C.c_int; L.telr_last_counters.argtypes = [vp, C.POINTER(Counters)]
    L.telr_last_dp_classes.restype = C.c_int; L.telr_last_dp_classes.argtypes = [vp, vp]
    # debug taps (not part of the public header; used by the stage-level parity tests)
    L.telr_debug_n_anchor.restype = i64; L.telr_debug_n_anchor.argtypes = [vp]
    L.telr_debug_dp_retries.restype = i64; L.telr_debug_dp_retries.argtypes = [vp]
    L.telr_debug_pk_launches.restype = i64; L.telr_debug_pk_launches.argtypes = [vp]
    L.telr_debug_fetch.restype = C.c_int; L.telr_debug_fetch.argtypes = [vp, cp, vp, i64]
    L.telr_debug_n_chain.restype = i64; L.telr_debug_n_chain.argtypes = [vp]
    L.telr_debug_chains.restype = vp; L.telr_debug_chains.argtypes = [vp]
    L.telr_debug_index.restype = C.c_int; L.telr_debug_index.argtypes = [vp, vp, vp, vp, vp]
    L.telr_debug_pack.restype = C.c_int; L.telr_debug_pack.argtypes = [cp, i32, C.c_int, vp, vp]
    L.telr_debug_mid_occ.restype = i32; L.telr_debug_mid_occ.argtypes = [vp, C.POINTER(MapOpt)]
    _lib = L
    return L
