"""Test double with the surface of telr_amd.aligner.Engine/Index, backed by the CPU oracle.
Lets the host pipeline (telr_te / telr_liftover / telr_af / locus_pipeline) run twice — once on the
HIP engine, once on the oracle — so that whole-pipeline outputs can be compared for equality."""
import numpy as np

from oracle import binding as ob
from telr_amd.aligner import MapResult


class _Targets(object):
    def __init__(self, seqs):
        self.len = np.array([len(s) for s in seqs], np.int32)


class OracleIndexAdapter(object):
    def __init__(self, seqs, io):
        if isinstance(seqs, tuple) and len(seqs) == 3:          # (byte buffer, offsets, lengths), as the engine's SeqSet takes it
            buf, off, ln = seqs
            seqs = [bytes(buf[int(o):int(o) + int(l)]) for o, l in zip(off, ln)]
        seqs = [s if isinstance(s, str) else bytes(s).decode() for s in seqs]
        self.ix = ob.OracleIndex(seqs, io)
        self.targets = _Targets(seqs)

    def map_raw(self, queries, mo, qtarget=None):
        q = [s if isinstance(s, str) else bytes(s).decode() for s in queries]
        return self.ix.map(q, mo, qtarget=qtarget)

    def free_raw(self, r):
        pass

    def result_arrays(self, r):
        return MapResult(r["alns"], r["cigars"])

    def map(self, queries, mo, qtarget=None):
        return self.result_arrays(self.map_raw(queries, mo, qtarget))

    def depth_medians(self, r, iv_tid, iv_s, iv_e):
        return ob.depth_medians(r["alns"], r["cigars"], self.targets.len, iv_tid, iv_s, iv_e)


class OracleBackend(object):
    def index(self, seqs, io):
        return OracleIndexAdapter(seqs, io)

    def seqset(self, seqs):
        return list(seqs)
