"""Effect of the chaining look-back (64 / 128 / 256 predecessors) on the records of the bench data set."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from telr_amd import synth
from telr_amd.aligner import Engine
from telr_amd.presets import preset

d = synth.make_stage1_dataset(seed=20261002, read_seed=20261002 + 1000)
eng = Engine(0)
io, mo = preset("map-ont")
ix = eng.index([bytes(d["ref"]).decode()], io)
qs = eng.seqset(d["reads"])
F = ["qid", "tid", "qs", "qe", "ts", "te", "flags"]
base = None
for H in (256, 128, 64):
    m = mo.copy(); m.chain_lookback = H
    ix.map(qs, m)
    t0 = time.time(); r = ix.map(qs, m); dt = time.time() - t0
    a = r.alns
    prim = a[(a["flags"] & 1) != 0]
    print("H=%d  %.1f ms  records %d primaries %d  sum dp_score %d  sum mlen %d  chain stage %.2f ms" % (H, dt * 1e3, len(a), len(prim), int(a["dp_score"].sum()), int(a["mlen"].sum()), eng.stage_ms().get("chain", 0)))
    if base is None:
        base = prim.copy(); continue
    if len(prim) == len(base):
        same = np.ones(len(prim), bool)
        for f in F:
            same &= prim[f] == base[f]
        ds = prim["dp_score"].astype(np.int64) - base["dp_score"]
        print("     primaries with identical coordinates %.5f, identical score %.5f, score sum diff %d (rel %.2e)" % (same.mean(), (ds == 0).mean(), int(ds.sum()), ds.sum() / base["dp_score"].sum()))
