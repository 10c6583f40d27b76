"""Object layer over the C ABI: Engine (context), SeqSet, Index, MapResult.

This is the in-process replacement of the reference's `subprocess.call(["minimap2", ...])`
boundary (TELR_alignment.py:69-82 and the five other sites listed in include/telr_hip.h).
"""
import ctypes as C
import numpy as np
from . import _lib
from ._abi import IdxOpt, MapOpt, Counters, ALN_DTYPE, N_STAGES, N_DPCLS
from .fasta import concat


def _np_from(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    dt = np.dtype(dtype)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt, count=n).copy()


_DEFERRED = None          # list of (free function, handle) while deferred_frees() is active


class deferred_frees:
    """`with deferred_frees():` -- sequence sets and indexes freed inside the block (explicitly or by the garbage collector) give
    their device memory back when the block ends.  hipFree waits for the whole device, so a free on one context stalls its host
    thread behind whatever another context of the process is running (the loci bundle: S4 / S5 / S7 next to S6)."""

    def __enter__(self):
        global _DEFERRED
        self.outer = _DEFERRED
        if _DEFERRED is None:
            _DEFERRED = []
        return self

    def __exit__(self, *exc):
        global _DEFERRED
        if self.outer is None:
            todo, _DEFERRED = _DEFERRED, None
            for fn, h in todo:
                fn(h)
        return False


class Engine:
    """One per process per device."""

    def __init__(self, device=0, background=False):
        self.L = _lib.lib()
        h = C.c_void_p()
        rc = (self.L.telr_init_background if background else self.L.telr_init)(device, C.byref(h))
        if rc != 0:
            raise _lib.TelrError("telr_init(%d): %s" % (device, self.L.telr_strerror(rc).decode()))
        self.h = h
        self.device = device

    def _chk(self, rc, what):
        if rc != 0:
            raise _lib.TelrError("%s: %s [%s]" % (what, self.L.telr_strerror(rc).decode(),
                                                  self.L.telr_last_error(self.h).decode()))

    def device_name(self):
        b = C.create_string_buffer(256)
        self.L.telr_device_name(self.h, b, 256)
        return b.value.decode()

    def seqset(self, seqs):
        return SeqSet(self, seqs)

    def index(self, targets, io):
        return Index(self, targets, io)

    def stage_ms(self):
        a = np.zeros(N_STAGES, np.float32)
        self.L.telr_stage_ms(self.h, a.ctypes.data)
        return {self.L.telr_stage_name(i).decode(): float(a[i]) for i in range(N_STAGES)}

    def counters(self):
        c = Counters()
        self.L.telr_last_counters(self.h, C.byref(c))
        return {k: getattr(c, k) for k, _ in Counters._fields_}

    def dp_classes(self):
        """per DP class: (problems, cells, steps, algorithmic bytes) of the last map call"""
        a = np.zeros(N_DPCLS * 4, np.int64)
        self.L.telr_last_dp_classes(self.h, a.ctypes.data)
        return a.reshape(N_DPCLS, 4)

    def release_scratch(self):
        """give the context's grow-only scratch back to the device (the next call allocates what it needs again)"""
        self._chk(self.L.telr_release_scratch(self.h), "telr_release_scratch")

    def mem_info(self):
        """(free, total) bytes of the device"""
        fr, tot = C.c_int64(), C.c_int64()
        self._chk(self.L.telr_device_mem(self.h, C.byref(fr), C.byref(tot)), "telr_device_mem")
        return fr.value, tot.value

    def worker(self):
        """A second context on the same device (created on first use, closed with this one): a host thread can run an engine
        call on it while this context runs another -- contexts are not re-entrant, different contexts are independent
        (the second range slot of a large call is such a context).  Sequence sets and indexes are plain device data and may be
        used from either."""
        w = getattr(self, "_worker", None)
        if w is None:
            w = self._worker = Engine(self.device, background=True)      # its kernels queue behind this context's
        return w

    def close(self):
        w = getattr(self, "_worker", None)
        if w is not None:
            self._worker = None
            w.close()
        if getattr(self, "h", None):
            self.L.telr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DevWords:
    """`n` int32 words of library-owned device memory, presented through the CUDA array interface (torch.as_tensor takes it
    without a copy); keeps the owning set alive"""
    def __init__(self, ptr, n, owner):
        self.owner = owner
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}


class SeqSet:
    def __init__(self, eng, seqs):
        self.eng = eng
        if isinstance(seqs, tuple) and len(seqs) == 3:
            buf, off, ln = seqs
        else:
            buf, off, ln = concat(seqs)
        self.len = np.ascontiguousarray(ln, dtype=np.int32)
        off = np.ascontiguousarray(off, dtype=np.int64)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        h = C.c_void_p()
        eng._chk(eng.L.telr_seqset_create(eng.h, len(self.len), buf.ctypes.data, off.ctypes.data, self.len.ctypes.data,
                                          C.byref(h)), "telr_seqset_create")
        self.h = h
        self.n = len(self.len)

    def bases(self):
        return int(self.eng.L.telr_seqset_bases(self.h))

    def subset(self, idx, eng=None, rc=None):
        """new set = copies of sequences idx (repeats allowed), gathered on the device from the packed form
        (eng: the context whose stream does the gather; default the one the set was made on); rc: one flag per copy, set = the copy
        is the reverse complement of its source (telr_seqset_subset_rc)"""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        eng = self.eng if eng is None else eng
        sub = SeqSet.__new__(SeqSet)
        sub.eng = eng
        h = C.c_void_p()
        if rc is None:
            eng._chk(eng.L.telr_seqset_subset(eng.h, self.h, len(idx), idx.ctypes.data, C.byref(h)), "telr_seqset_subset")
        else:
            rc = np.ascontiguousarray(rc, dtype=np.uint8)
            if len(rc) != len(idx):
                raise ValueError("subset: one rc flag per index")
            eng._chk(eng.L.telr_seqset_subset_rc(eng.h, self.h, len(idx), idx.ctypes.data, rc.ctypes.data, C.byref(h)), "telr_seqset_subset_rc")
        sub.h = h; sub.len = self.len[idx].copy(); sub.n = len(idx)
        return sub

    def packed(self):
        """the set's two word arrays as torch int32 tensors ON THE DEVICE, zero-copy views of the library's memory (valid while
        the set lives): what the N > 1 hand-offs put on the wire (telr_seqset_packed)"""
        import torch
        if not torch.cuda.is_available():
            raise _lib.TelrError("SeqSet.packed: torch sees no GPU in this process -- import torch BEFORE the first telr_amd call "
                                 "(the torch wheel carries its own HIP runtime; a process initialises only one)")
        p2, pn, n2, nn = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_int64()
        self.eng._chk(self.eng.L.telr_seqset_packed(self.h, C.byref(p2), C.byref(pn), C.byref(n2), C.byref(nn)), "telr_seqset_packed")
        dev = "cuda:%d" % self.eng.device

        def view(ptr, n):
            if n == 0:
                return torch.zeros(0, dtype=torch.int32, device=dev)
            return torch.as_tensor(_DevWords(ptr, n, self), device=dev)
        return view(p2.value, n2.value), view(pn.value, nn.value)

    @classmethod
    def from_packed(cls, eng, lengths, seq2, nmask):
        """a set built from packed words that are already on the device (torch int32 tensors, e.g. what an all-to-all delivered):
        one device-to-device copy, nothing is unpacked (telr_seqset_from_packed)"""
        import torch
        s = cls.__new__(cls)
        s.eng = eng
        s.len = np.ascontiguousarray(lengths, dtype=np.int32)
        seq2 = seq2.contiguous(); nmask = nmask.contiguous()
        torch.cuda.current_stream(seq2.device).synchronize()      # the words were produced on torch's stream (RCCL), the copy runs on the context's
        h = C.c_void_p()
        eng._chk(eng.L.telr_seqset_from_packed(eng.h, len(s.len), s.len.ctypes.data, C.c_void_p(seq2.data_ptr() if seq2.numel() else 0), int(seq2.numel()),
                                               C.c_void_p(nmask.data_ptr() if nmask.numel() else 0), int(nmask.numel()), C.byref(h)), "telr_seqset_from_packed")
        s.h = h; s.n = len(s.len)
        return s

    def free(self):
        if getattr(self, "h", None):
            if _DEFERRED is not None:               # hipFree waits for the device: not while another context's call is running on it
                _DEFERRED.append((self.eng.L.telr_seqset_free, self.h))
            else:
                self.eng.L.telr_seqset_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MapResult:
    def __init__(self, alns, cigars):
        self.alns = alns
        self.cigars = cigars

    def cigar(self, i):
        a = self.alns[i]
        return self.cigars[a["cigar_off"]:a["cigar_off"] + a["n_cigar"]]

    def cigar_string(self, i):
        return "".join("%d%s" % (c >> 4, "MID"[c & 0xf]) for c in self.cigar(i))


class Index:
    def __init__(self, eng, targets, io):
        self.eng = eng
        self.targets = targets if isinstance(targets, SeqSet) else SeqSet(eng, targets)
        self.io = io
        h = C.c_void_p()
        eng._chk(eng.L.telr_index_build(eng.h, self.targets.h, C.byref(io), C.byref(h)), "telr_index_build")
        self.h = h

    def stats(self):
        a, b = C.c_int64(0), C.c_int64(0)
        self.eng.L.telr_index_stats(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def map_raw(self, queries, mo, qtarget=None):
        """-> opaque result handle (caller frees with free_raw)."""
        q = queries if isinstance(queries, SeqSet) else SeqSet(self.eng, queries)
        qt = None if qtarget is None else np.ascontiguousarray(qtarget, dtype=np.int32)
        r = C.c_void_p()
        self.eng._chk(self.eng.L.telr_map(self.eng.h, self.h, q.h, None if qt is None else qt.ctypes.data,
                                          C.byref(mo), C.byref(r)), "telr_map")
        return r

    def result_from_arrays(self, alns, cigars):
        """raw result handle over caller-held records + CIGAR words (copied): for the writers on records mapped elsewhere"""
        alns = np.ascontiguousarray(alns, dtype=ALN_DTYPE); cigars = np.ascontiguousarray(cigars, dtype=np.uint32)
        r = C.c_void_p()
        self.eng._chk(self.eng.L.telr_result_from_arrays(self.eng.h, alns.ctypes.data, len(alns), cigars.ctypes.data, len(cigars), C.byref(r)), "telr_result_from_arrays")
        return r

    def result_from_device_cigars(self, alns, cigars_t):
        """the same over CIGAR words that are on the device already (a torch int32 / uint8 tensor, e.g. what an all-to-all
        delivered): they become the result's device copy as they are (telr_result_from_device_cigars)"""
        import torch
        alns = np.ascontiguousarray(alns, dtype=ALN_DTYPE)
        t = cigars_t.contiguous()
        n = t.numel() * t.element_size() // 4
        torch.cuda.current_stream(t.device).synchronize()
        r = C.c_void_p()
        self.eng._chk(self.eng.L.telr_result_from_device_cigars(self.eng.h, alns.ctypes.data, len(alns), C.c_void_p(t.data_ptr() if n else 0), n, C.byref(r)),
                      "telr_result_from_device_cigars")
        return r

    def free_raw(self, r):
        self.eng.L.telr_result_free(r)

    # ---- one BAM written by N ranks: the slice of this rank (include/telr_hip.h: telr_write_bam_slice ...) ----
    def write_bam_slice(self, r, queries, qnames, tnames, emit, with_header, md=True, cs=True, softclip=True, rg=None, cmdline="telr_map", level=1, unmapped=True):
        """-> segment handle: the BGZF blocks of the records with emit[i] != 0 (+ the unmapped reads of `queries` when `unmapped`)
        as an image on the device; the other records only feed the SA tags"""
        qa, ta = self._cstr_array(qnames), self._cstr_array(tnames)
        flags = (1 if md else 0) | (2 if cs else 0) | (4 if softclip else 0) | (0 if unmapped else 8)
        rg_id, rg_sm, rg_lb = (None, None, None) if rg is None else tuple(x.encode() for x in rg)
        emit = np.ascontiguousarray(emit, dtype=np.uint8)
        seg = C.c_void_p()
        self.eng._chk(self.eng.L.telr_write_bam_slice(self.eng.h, r, queries.h, self.h, qa, ta, flags, rg_id, rg_sm, rg_lb, cmdline.encode(),
                                                      emit.ctypes.data, 1 if with_header else 0, level, C.byref(seg)), "telr_write_bam_slice")
        return seg

    def segment_info(self, seg):
        out = np.zeros(4, np.int64)
        self.eng._chk(self.eng.L.telr_bam_segment_info(seg, out.ctypes.data), "telr_bam_segment_info")
        return dict(zip(("bytes", "mapped_records", "unmapped_reads", "uncompressed_bytes"), (int(x) for x in out)))

    def segment_entries(self, seg, file_off):
        """-> (tid, start, end, virtual offset) arrays of the slice's mapped records in file order, and the virtual offset behind them"""
        n = self.segment_info(seg)["mapped_records"]
        tid, ts, te = (np.zeros(n, np.int32) for _ in range(3)); vb = np.zeros(n, np.uint64); v_end = C.c_uint64()
        self.eng._chk(self.eng.L.telr_bam_segment_entries(seg, int(file_off), tid.ctypes.data, ts.ctypes.data, te.ctypes.data, vb.ctypes.data, C.byref(v_end)), "telr_bam_segment_entries")
        return tid, ts, te, vb, int(v_end.value)

    def segment_write(self, seg, path, file_off, is_last):
        self.eng._chk(self.eng.L.telr_bam_segment_write(self.eng.h, seg, path.encode(), int(file_off), 1 if is_last else 0), "telr_bam_segment_write")

    def segment_free(self, seg):
        self.eng.L.telr_bam_segment_free(seg)

    def bai_write(self, path, tid, ts, te, vb, v_end, n_unmapped, tlens):
        tid, ts, te = (np.ascontiguousarray(x, np.int32) for x in (tid, ts, te)); vb = np.ascontiguousarray(vb, np.uint64); tl = np.ascontiguousarray(tlens, np.int32)
        self.eng._chk(self.eng.L.telr_bai_write(path.encode(), len(tid), tid.ctypes.data, ts.ctypes.data, te.ctypes.data, vb.ctypes.data, int(v_end), int(n_unmapped), len(tl), tl.ctypes.data), "telr_bai_write")

    def result_arrays(self, r):
        L = self.eng.L
        alns = _np_from(L.telr_result_alns(r), L.telr_result_count(r), ALN_DTYPE)
        cig = _np_from(L.telr_result_cigars(r), L.telr_result_cigar_count(r), np.uint32)
        return MapResult(alns, cig)

    def map(self, queries, mo, qtarget=None):
        r = self.map_raw(queries, mo, qtarget)
        try:
            return self.result_arrays(r)
        finally:
            self.free_raw(r)

    # ---- text emitters (the reference's own boundary: PAF / SAM files) -----------------------
    @staticmethod
    def _cstr_array(names):
        if isinstance(names, C.Array):          # already built (a caller that writes many files with the same names)
            return names
        arr = (C.c_char_p * max(1, len(names)))()
        for i, n in enumerate(names):
            arr[i] = n.encode() if isinstance(n, str) else bytes(n)
        return arr

    def write_paf(self, r, qnames, tnames, path, with_cigar=True, append=False):
        qa, ta = self._cstr_array(qnames), self._cstr_array(tnames)
        self.eng._chk(self.eng.L.telr_write_paf(r, qa, ta, 1 if with_cigar else 0, path.encode(), 1 if append else 0), "telr_write_paf")

    def write_sam(self, r, qnames, queries, tnames, targets, path, md=True, cs=True, softclip=True, rg=None, cmdline="telr_map",
                  primary_only=False, coordinate_sorted=False, header=True, unmapped=True):
        """queries / targets: lists of sequences (str) or (buf, off, len) triples as given to seqset().
        primary_only + coordinate_sorted + header=False = the text of `samtools view -F0x900 sorted.bam`
        (the polishing hand-off to wtpoa-cns, TELR_assembly.py:208,228)."""
        qb, qo, ql = queries if isinstance(queries, tuple) else concat(queries)
        tb, to, tl = targets if isinstance(targets, tuple) else concat(targets)
        qb = np.ascontiguousarray(qb, np.uint8); tb = np.ascontiguousarray(tb, np.uint8)
        qo = np.ascontiguousarray(qo, np.int64); to = np.ascontiguousarray(to, np.int64)
        ql = np.ascontiguousarray(ql, np.int32); tl = np.ascontiguousarray(tl, np.int32)
        qa, ta = self._cstr_array(qnames), self._cstr_array(tnames)
        flags = (1 if md else 0) | (2 if cs else 0) | (4 if softclip else 0) | (0 if unmapped else 8) | (16 if primary_only else 0) | \
            (32 if coordinate_sorted else 0) | (0 if header else 64)
        rg_id, rg_sm, rg_lb = (None, None, None) if rg is None else tuple(x.encode() for x in rg)
        self.eng._chk(self.eng.L.telr_write_sam(r, len(ql), qa, qb.ctypes.data, qo.ctypes.data, ql.ctypes.data, len(tl), ta,
                                                tb.ctypes.data, to.ctypes.data, tl.ctypes.data, flags, rg_id, rg_sm, rg_lb,
                                                cmdline.encode(), path.encode()), "telr_write_sam")

    def write_bam(self, r, qnames, queries, tnames, targets, path, md=True, cs=True, softclip=True, rg=None, cmdline="telr_map",
                  index=True, level=1):
        """coordinate-sorted BAM + .bai (samtools sort + index, TELR_alignment.py:103-114)"""
        qb, qo, ql = queries if isinstance(queries, tuple) else concat(queries)
        tb, to, tl = targets if isinstance(targets, tuple) else concat(targets)
        qb = np.ascontiguousarray(qb, np.uint8); tb = np.ascontiguousarray(tb, np.uint8)
        qo = np.ascontiguousarray(qo, np.int64); to = np.ascontiguousarray(to, np.int64)
        ql = np.ascontiguousarray(ql, np.int32); tl = np.ascontiguousarray(tl, np.int32)
        qa, ta = self._cstr_array(qnames), self._cstr_array(tnames)
        flags = (1 if md else 0) | (2 if cs else 0) | (4 if softclip else 0)
        rg_id, rg_sm, rg_lb = (None, None, None) if rg is None else tuple(x.encode() for x in rg)
        self.eng._chk(self.eng.L.telr_write_bam(r, len(ql), qa, qb.ctypes.data, qo.ctypes.data, ql.ctypes.data, len(tl), ta,
                                                tb.ctypes.data, to.ctypes.data, tl.ctypes.data, flags, rg_id, rg_sm, rg_lb,
                                                cmdline.encode(), path.encode(), 1 if index else 0, level), "telr_write_bam")

    def write_bam_device(self, r, queries, qnames, tnames, path, md=True, cs=True, softclip=True, rg=None, cmdline="telr_map",
                         index=True, level=1, unmapped=True):
        """the same file built on the device from the resident reads / reference / CIGARs (telr_write_bam_dev);
        queries = the SeqSet the result was mapped from"""
        qa, ta = self._cstr_array(qnames), self._cstr_array(tnames)
        flags = (1 if md else 0) | (2 if cs else 0) | (4 if softclip else 0) | (0 if unmapped else 8)
        rg_id, rg_sm, rg_lb = (None, None, None) if rg is None else tuple(x.encode() for x in rg)
        self.eng._chk(self.eng.L.telr_write_bam_dev(self.eng.h, r, queries.h, self.h, qa, ta, flags, rg_id, rg_sm, rg_lb,
                                                    cmdline.encode(), path.encode(), 1 if index else 0, level), "telr_write_bam_dev")

    def bam_prepare(self, path, est_bytes):
        """start creating the output file in the background (call before map_raw; see telr_bam_prepare)"""
        self.eng._chk(self.eng.L.telr_bam_prepare(self.eng.h, path.encode(), int(est_bytes)), "telr_bam_prepare")

    def bam_discard(self):
        """drop a prepared output file that will not be written (telr_bam_discard; the caller unlinks the file)"""
        if getattr(self.eng, "h", None):
            self.eng.L.telr_bam_discard(self.eng.h)

    def bam_release_wait(self):
        """wait until the mappings of earlier output files are taken apart (telr_bam_release_wait)"""
        self.eng.L.telr_bam_release_wait()

    def bam_stage_ms(self):
        a = np.zeros(8, np.float32); b = np.zeros(12, np.float32)
        self.eng.L.telr_debug_bam_ms(a.ctypes.data); self.eng.L.telr_debug_bam_sink_ms(b.ctypes.data)
        d = dict(zip(("upload", "scan_size", "sort_offsets", "write_records", "bgzf", "d2h_file", "bai_host_overlapped", "total"), (float(x) for x in a)))
        d.update(zip(("sink_allocate_bg", "sink_map_bg", "sink_wait", "sink_mapping_used", "stream_wait_coder", "stream_wait_dma", "stream_host_copy", "stream_wait_slot", "stream_loop", "sink_stop", "sink_truncate"), (float(x) for x in b)))
        return d

    def consensus(self, r, queries, min_depth=3, poa=False):
        """consensus of this index's targets from the primary records of raw result r -> list of str: the pile-up vote
        (telr_consensus_build) or, poa=True, the window partial-order consensus (telr_poa_build)"""
        h = C.c_void_p()
        fn, what = (self.eng.L.telr_poa_build, "telr_poa_build") if poa else (self.eng.L.telr_consensus_build, "telr_consensus_build")
        self.eng._chk(fn(self.eng.h, r, queries.h, self.h, int(min_depth), C.byref(h)), what)
        try:
            L = self.eng.L
            n = int(L.telr_consensus_count(h))
            off = _np_from(L.telr_consensus_off(h), n, np.int64); ln = _np_from(L.telr_consensus_len(h), n, np.int32)
            tot = int(off[-1] + ln[-1]) if n else 0
            buf = _np_from(L.telr_consensus_seq(h), tot, np.uint8)
            return [bytes(buf[off[i]:off[i] + ln[i]]).decode() for i in range(n)]
        finally:
            self.eng.L.telr_consensus_free(h)

    def depth_medians(self, r, iv_tid, iv_s, iv_e):
        """Medians over 0-based inclusive intervals, from a raw result handle."""
        tl = self.targets.len
        a, b, c = (np.ascontiguousarray(x, dtype=np.int32) for x in (iv_tid, iv_s, iv_e))
        out = np.zeros(len(a), np.float64)
        self.eng._chk(self.eng.L.telr_depth_medians(self.eng.h, r, len(tl), tl.ctypes.data, len(a), a.ctypes.data,
                                                    b.ctypes.data, c.ctypes.data, out.ctypes.data), "telr_depth_medians")
        return out

    # ---- debug taps for the stage-level parity tests ----------------------------------
    def debug_dump(self):
        L = self.eng.L
        n_mz, n_ent = self.stats()
        eh = np.zeros(n_ent, np.uint64); eo = np.zeros(n_ent + 1, np.uint32); pos = np.zeros(n_mz, np.uint32)
        self.eng._chk(L.telr_debug_index(self.eng.h, self.h, eh.ctypes.data, eo.ctypes.data, pos.ctypes.data), "debug_index")
        return eh, eo, pos

    def debug_mid_occ(self, mo):
        return int(self.eng.L.telr_debug_mid_occ(self.h, C.byref(mo)))

    def debug_last_batch(self, nq):
        L = self.eng.L
        na = int(L.telr_debug_n_anchor(self.eng.h))
        out = {}
        for name, dt, n in (("skeys", np.uint64, na), ("chain_f", np.int32, na), ("chain_p", np.int32, na), ("q_aoff", np.int32, nq + 1)):
            a = np.zeros(n, dt)
            if n:
                self.eng._chk(L.telr_debug_fetch(self.eng.h, name.encode(), a.ctypes.data, a.nbytes), "debug_fetch " + name)
            out[name] = a
        nc = int(L.telr_debug_n_chain(self.eng.h))
        out["chains"] = _np_from(L.telr_debug_chains(self.eng.h), nc * 9, np.int32).reshape(-1, 9)
        return out

    def free(self):
        if getattr(self, "h", None):
            if _DEFERRED is not None:
                _DEFERRED.append((self.eng.L.telr_index_free, self.h))
            else:
                self.eng.L.telr_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
