#!/usr/bin/env python3
"""Which spiked insertions does the per-locus bundle NOT recover, and why?  (configs[1] data set, GPU box)
Prints one line per lost locus: truth, what the liftover reported, the annotation rows of its contig."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from telr_amd import synth, locus_pipeline, telr_assembly
from telr_amd.aligner import Engine
from telr_amd.presets import preset

d = synth.make_stage1_dataset(seed=20261002, read_seed=20261002 + 1000)
eng = Engine(0)
io, mo = preset("map-ont")
ref = bytes(d["ref"]).decode()
ix = eng.index([ref], io)
qs = eng.seqset(d["reads"])
res = ix.map(qs, mo)
loci = synth.make_loci_from_dataset(d, 200)
mode = sys.argv[1] if len(sys.argv) > 1 else "engine"
if mode == "engine":
    wr = telr_assembly.window_reads(res.alns, {"chr2L": 0}, [("chr2L", l["truth"]["pos"], l["truth"]["pos"] + 1) for l in loci])
    for l, idx in zip(loci, wr):
        l["read_idx"] = idx.astype(np.int32)
io10, _ = preset("asm10")
ix10 = eng.index([ref], io10)
lib_names = ["fam%d" % i for i in range(len(d["library"]))]
lib = [bytes(x).decode() for x in d["library"]]
out = locus_pipeline.run_loci(eng, ix10, ["chr2L"], lambda ch: ref, loci, lib_names, lib, read_set=qs)
by = {}
for r in out["liftover"]:
    by.setdefault(locus_pipeline.locus_of_report(r), []).append(r)
ann = {}
for a in out["annotation"]:
    ann.setdefault(a[0], []).append(a)
lost = 0
for l in loci:
    t = l["truth"]
    rs = by.get(l["name"], [])
    ok = any(r["report"]["type"] == "non-reference" and abs(r["report"]["start"] - t["pos"]) <= 20 and r["report"]["strand"] == t["strand"] and r["report"]["family"] == t["family"] for r in rs)
    if ok:
        continue
    lost += 1
    print("LOST", l["name"], "truth", t, "te_len", len(d["library"][int(t["family"][3:])]), "contig_len", len(l["contig"]), "alt_len", len(l["alt"]))
    print("   annotation:", ann.get(l["name"]))
    for r in rs:
        rep = r["report"]
        print("   report:", {k: rep.get(k) for k in ("type", "chrom", "start", "end", "strand", "family", "gap", "TSD_length", "comment")}, "num_hits", r.get("num_hits"))
print("lost", lost, "of", len(loci), "mode", mode)
