"""60 blocking telr_map calls on the bench dataset: step-time spread and device-memory growth (none)."""
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import torch, ctypes as C
from telr_amd import synth
from telr_amd.aligner import Engine
from telr_amd.presets import preset
io, mo = preset("map-ont")
d = synth.make_stage1_dataset(genome_len=23513712, n_reads=10000, total_bases=470_000_000, seed=20261002, n_ins=200, read_seed=20261002 + 1000)
e = Engine(0); ix = e.index([bytes(d["ref"]).decode()], io); qs = e.seqset(d["reads"])
def one():
    r = C.c_void_p(); assert e.L.telr_map(e.h, ix.h, qs.h, None, C.byref(mo), C.byref(r)) == 0
    e.L.telr_result_wait(r); n = e.L.telr_result_count(r); e.L.telr_result_free(r); return n
one(); one()
free0 = torch.cuda.mem_get_info(0)[0]
ts = []
for i in range(60):
    t0 = time.time(); n = one(); ts.append((time.time() - t0) * 1e3)
free1 = torch.cuda.mem_get_info(0)[0]
import statistics
print(json.dumps({"records": n, "ms_min": min(ts), "ms_median": statistics.median(ts), "ms_max": max(ts), "free_before_GB": free0 / 1e9, "free_after_GB": free1 / 1e9}))
