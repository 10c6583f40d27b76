"""How much do alignments change with the first-pass band factor (fill_band_q4)?  Maps the bench data set with several
values and compares every record with the q4 = 8 result (same records, score / match-count differences)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from telr_amd import synth
from telr_amd.aligner import Engine
from telr_amd.presets import preset

PRESET = os.environ.get("BAND_PRESET", "map-ont")
ERR = (0.013, 0.065, 0.052) if PRESET == "map-pb" else (0.04, 0.02, 0.04)       # CLR-like 13 % (1:5:4) / ONT-like 10 % (4:2:4)
d = synth.make_stage1_dataset(seed=20261002, read_seed=20261002 + 1000, err=ERR)
eng = Engine(0)
io, mo = preset(PRESET)
ix = eng.index([bytes(d["ref"]).decode()], io)
qs = eng.seqset(d["reads"])
F = ["qid", "tid", "qs", "qe", "ts", "te", "flags"]
base = None
import time
for q4 in (8, 6, 5, 4, 3):
    m = mo.copy(); m.fill_band_q4 = q4
    ix.map(qs, m)
    t0 = time.time(); r = ix.map(qs, m); dt = (time.time() - t0) * 1e3
    print('  q4=%d: %.1f ms (blocking call), retries %d, dp stage %.1f ms' % (q4, dt, int(eng.L.telr_debug_dp_retries(eng.h)), eng.stage_ms().get('dp', 0)))
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
