import sys, os, time, threading, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from telr_amd import synth
from telr_amd.aligner import Engine, _np_from
from telr_amd._abi import ALN_DTYPE
from telr_amd.presets import preset
NF = int(sys.argv[1]); K = int(sys.argv[2])
io, mo = preset("map-ont")
d = synth.make_stage1_dataset(genome_len=23513712, n_reads=10000, total_bases=470_000_000, seed=20261002, n_ins=200, read_seed=20261002 + 1000)
ref = bytes(d["ref"]).decode()
engs = [Engine(0) for _ in range(NF)]
ix = engs[0].index([ref], io)
qs = engs[0].seqset(d["reads"])
nb = qs.bases()
import ctypes as C
def run(e, n, out):
    held = []
    for _ in range(n):
        r = C.c_void_p()
        rc = e.L.telr_map(e.h, ix.h, qs.h, None, C.byref(mo), C.byref(r))
        assert rc == 0, rc
        cnt = e.L.telr_result_count(r)
        while held: e.L.telr_result_free(held.pop())
        held.append(r)
    for r in held:
        e.L.telr_result_wait(r); e.L.telr_result_free(r)
    out.append(n)
for e in engs: run(e, 2, [])
torch.cuda.synchronize(0)
t0 = time.time()
outs = []
ths = [threading.Thread(target=run, args=(e, K // NF, outs)) for e in engs]
for t in ths: t.start()
for t in ths: t.join()
torch.cuda.synchronize(0)
dt = time.time() - t0
n = sum(outs)
print(json.dumps({"inflight": NF, "steps": n, "ms_per_step": dt / n * 1e3, "gbp_s": nb * n / dt / 1e9}))
