import sys, os, time, json, numpy as np
sys.path.insert(0, os.getcwd())
from telr_amd import synth, shard
from telr_amd.aligner import Engine
from telr_amd.presets import preset
io, mo = preset("map-ont")
d = synth.make_stage1_dataset(genome_len=23513712, n_reads=10000, total_bases=470_000_000, seed=20261002, n_ins=200, read_seed=20261002 + 1000)
e = Engine(0); ix = e.index([bytes(d["ref"]).decode()], io); qs = e.seqset(d["reads"])
for _ in range(3):
    r = ix.map_raw(qs, mo); ix.free_raw(r)
tm, tf = [], []
for _ in range(20):
    t0 = time.time(); r = ix.map_raw(qs, mo); t1 = time.time(); ix.free_raw(r); t2 = time.time()
    tm.append((t1 - t0) * 1e3); tf.append((t2 - t1) * 1e3)
print(json.dumps({"map_raw_ms": float(np.median(tm)), "free_raw_ms": float(np.median(tf)), "map_wall_stage_ms": e.stage_ms().get("map_wall")}))
