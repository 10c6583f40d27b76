"""Iteration aid: DP classes (problems, cells) and stage times of the polishing map (`-ax P`-style: reads against their own draft contig,
qtarget) on simulated loci, bw 2000 against bw 500.  usage (through gpurun): python3 tools/polish_classes.py"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from telr_amd.aligner import Engine
from telr_amd.presets import preset
from telr_amd import synth
rng = np.random.default_rng(5)
drafts, reads = [], []
for k in range(200):
    L = 25000
    truth = synth.random_seq(rng, L)
    drafts.append(bytes(synth.mutate(rng, truth, 0.005, 0.003, 0.003)).decode())
    rs = []
    for _ in range(40):
        s = int(rng.integers(0, L - 9000)); r = synth.mutate(rng, truth[s:s + int(rng.integers(6000, 9000))], 0.04, 0.03, 0.03)
        rs.append(bytes(synth.revcomp_arr(r) if rng.integers(0, 2) else r).decode())
    reads.append(rs)
eng = Engine(0)
flat = [r for rs in reads for r in rs]
qt = np.array([k for k, rs in enumerate(reads) for _ in rs], np.int32)
qset = eng.seqset(flat)
import time
for bw in (2000, 500):
    io, mo = preset("map-ont"); mo.bw = bw
    ix = eng.index(drafts, io)
    for rep in range(2):
        t0 = time.time(); r = ix.map_raw(qset, mo, qtarget=qt); res = ix.result_arrays(r); dt = time.time() - t0
        ix.free_raw(r)
    dc = eng.dp_classes()
    print("bw", bw, "map %.1f ms" % (dt * 1e3), "bases", sum(len(x) for x in flat), {k: round(v, 1) for k, v in eng.stage_ms().items() if v > 0.5})
    print("  classes (problems, Mcells):", ", ".join("%d: %d %.0f" % (c, v[0], v[1] / 1e6) for c, v in enumerate(dc.tolist()) if v[0]))
    ix.free()
