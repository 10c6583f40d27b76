"""How much do alignments change with the first-pass band factor (fill_band_q4)?  Maps the bench data set with several
values and compares every record with the q4 = 8 result (same records, score / match-count differences)."""
import sys, os
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
for q4 in (8, 6, 5, 4, 3):
    m = mo.copy(); m.fill_band_q4 = q4
    r = ix.map(qs, m)
    a = r.alns
    if base is None:
        base = a.copy(); print("q4=8 records", len(a), "sum dp_score", int(a["dp_score"].sum()), "sum mlen", int(a["mlen"].sum())); continue
    same_n = len(a) == len(base)
    if same_n:
        same_pos = np.ones(len(a), bool)
        for f in F:
            same_pos &= a[f] == base[f]
        ds = a["dp_score"].astype(np.int64) - base["dp_score"]
        dm = a["mlen"].astype(np.int64) - base["mlen"]
        print("q4=%d records %d same-coords %.5f  score: equal %.5f lower %d higher %d sum-diff %d (rel %.2e)  mlen sum-diff %d"
              % (q4, len(a), same_pos.mean(), (ds == 0).mean(), int((ds < 0).sum()), int((ds > 0).sum()), int(ds.sum()), ds.sum() / base["dp_score"].sum(), int(dm.sum())))
    else:
        print("q4=%d records %d (differs from %d)" % (q4, len(a), len(base)))
