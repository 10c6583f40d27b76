"""Randomised parity of the window partial-order consensus (telr_poa_build / k_poa_window against the oracle's tor_poa, string for
string): random contig lengths (shorter than a window, a few bases past a window border), depths from 1 to beyond the 64-piece cap,
error mixes from clean to 15 %, reads with N runs, long indels under and over the 30-base rule, both strands, several min_depth.
`python tests/fuzz_poa.py 60 <seed>` runs the long version."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob                      # noqa: E402
from telr_amd import synth                            # noqa: E402
from telr_amd.presets import preset                   # noqa: E402


def draw_case(seed):
    rng = np.random.default_rng(seed)
    n_loci = int(rng.integers(1, 5))
    sub, ins, dele = [float(x) for x in rng.choice([0.0, 0.01, 0.03, 0.06], 3)]
    drafts, reads = [], []
    for k in range(n_loci):
        L = int(rng.choice([150, 199, 200, 201, 405, 999, 1600, 2600, int(rng.integers(300, 4000))]))
        truth = synth.random_seq(rng, L)
        draft = synth.mutate(rng, truth, 0.01, 0.005, 0.005) if rng.random() < 0.8 else truth.copy()
        if rng.random() < 0.2 and len(draft) > 60:
            draft = draft.copy(); draft[20:23] = ord("N"); draft[40:50] |= 32
        depth = int(rng.choice([0, 1, 2, 3, 8, 20, 40, 80]))
        rs = []
        for _ in range(depth):
            t = truth
            if rng.random() < 0.3 and L > 400:         # a long difference: under (votes) or over (does not vote) the 30-base rule
                pos = int(rng.integers(100, L - 100)); ln = int(rng.choice([12, 25, 30, 31, 45]))
                t = np.concatenate([t[:pos], synth.random_seq(rng, ln), t[pos:]]) if rng.random() < 0.5 else np.concatenate([t[:pos], t[pos + ln:]])
            if L > 1200 and rng.random() < 0.5:
                s = int(rng.integers(0, len(t) - 1000)); t = t[s:s + int(rng.integers(900, len(t) - s + 1))]
            r = synth.mutate(rng, t, sub, ins, dele)
            if rng.random() < 0.1 and len(r) > 40:
                r = r.copy(); r[30:33] = ord("N")
            rs.append(bytes(synth.revcomp_arr(r) if rng.integers(0, 2) else r).decode())
        drafts.append(bytes(draft).decode()); reads.append(rs)
    return drafts, reads, str(rng.choice(["map-ont", "map-pb"]))


def run(engine, n_iter, seed0):
    for it in range(n_iter):
        seed = seed0 * 1000 + it
        drafts, reads, pname = draw_case(seed)
        flat = [r for rs in reads for r in rs]
        if not flat:
            continue
        io, mo = preset(pname); mo.bw = 2000
        qt = np.array([k for k, rs in enumerate(reads) for _ in rs], np.int32)
        ix = engine.index(drafts, io)
        qset = engine.seqset(flat)
        r = ix.map_raw(qset, mo, qtarget=qt)
        try:
            res = ix.result_arrays(r)
            for md in (3, 1):
                got = ix.consensus(r, qset, min_depth=md, poa=True)
                want = ob.consensus(res.alns, res.cigars, flat, drafts, min_depth=md, poa=True)
                if got != want:
                    bad = [(i, len(drafts[i]), len(reads[i]), len(got[i]), len(want[i])) for i in range(len(got)) if got[i] != want[i]]
                    raise AssertionError("POA differs: seed %d preset %s min_depth %d: (locus, draft length, reads, got, want) %s" % (seed, pname, md, bad))
        finally:
            ix.free_raw(r); ix.free(); qset.free()


if __name__ == "__main__":
    from telr_amd.aligner import Engine
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    run(Engine(0), n_iter, seed0)
    print("poa fuzz ok:", n_iter, "iterations in %.1f s" % (time.time() - t0))
