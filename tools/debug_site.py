"""debug: records of the window reads of one configs[1] site (usage: python tools/debug_site.py <locus name>)"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from telr_amd import synth, telr_assembly
from telr_amd.aligner import Engine
from telr_amd.presets import preset
name = sys.argv[1]
eng = Engine(0)
d = synth.make_stage1_dataset(seed=20261002, read_seed=20261002 + 1000)
loci = synth.make_loci_from_dataset(d, 200, reads_cap=10 ** 9)
l = [x for x in loci if x["name"] == name][0]
p = l["truth"]["pos"]; te_len = len(d["library"][int(l["truth"]["family"][3:])])
print(l["truth"], "te_len", te_len, "te copies near:", [(s, e) for s, e in d["te_copies"] if s < p + 9000 and e > p - 9000])
for pname in ("map-ont",):
  for lj in (20000, 0):
    io, mo = preset(pname); mo.bw_long = lj
    ix = eng.index([bytes(d["ref"]).decode()], io)
    res = ix.map(eng.seqset(d["reads"]), mo)
    wr = telr_assembly.window_reads(res.alns, {"chr2L": 0}, [("chr2L", p, p + 1)])[0]
    print("bw_long", lj, "window reads", len(wr))
    for q in wr.tolist():
        for i in np.nonzero(res.alns["qid"] == q)[0]:
            a = res.alns[i]
            if a["flags"] & 2: continue
            ops = res.cigar(i)
            big = [(("MID"[c & 15]), int(c >> 4)) for c in ops if (c & 15) in (1, 2) and (c >> 4) >= 200]
            t = int(a["ts"]); pos = []
            for c in ops:
                if (c & 15) in (1, 2) and (c >> 4) >= 200: pos.append(t - p)
                if (c & 15) != 1: t += int(c >> 4)
            print("  read", q, "len", a["qlen"], "q", a["qs"], a["qe"], "t-p", a["ts"] - p, a["te"] - p, "rev", (a["flags"] >> 3) & 1, "fl", a["flags"] & 7, "big", big, "at", pos)
