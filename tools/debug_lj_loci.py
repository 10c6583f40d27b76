"""debug: which loci of the configs[1] bundle change with the long join in the per-locus calls"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from telr_amd import synth, telr_assembly, locus_pipeline, presets as P
from telr_amd.aligner import Engine
eng = Engine(0)
d = synth.make_stage1_dataset(seed=20261002, read_seed=20261002 + 1000)
ref = bytes(d["ref"]).decode()
loci = synth.make_loci_from_dataset(d, 200, reads_cap=10 ** 9)
io10, _ = P.preset("asm10"); ix10 = eng.index([ref], io10)
lib_names = ["fam%d" % i for i in range(len(d["library"]))]; lib = [bytes(x).decode() for x in d["library"]]
qs = eng.seqset(d["reads"])
orig = P.preset
def run(lj):
    def pr(name):
        io, mo = orig(name)
        if name in ("map-ont", "map-pb"): mo.bw_long = lj
        return io, mo
    P.preset = pr
    import telr_amd.telr_te as T, telr_amd.telr_af as A, telr_amd.locus_pipeline as LP
    for m in (T, A, LP):
        if hasattr(m, "preset"): m.preset = pr
    L = [dict(l, read_idx=np.asarray(l["read_idx"], np.int32)) for l in loci]
    for l in L: l.pop("reads", None)
    return locus_pipeline.run_loci(eng, ix10, ["chr2L"], lambda ch: ref, L, lib_names, lib, read_set=qs)
a, b = run(0), run(20000)
def ok(out):
    by = {}
    for r in out["liftover"]: by.setdefault(locus_pipeline.locus_of_report(r), []).append(r["report"])
    s = set()
    for l in loci:
        t = l["truth"]
        if any(r["type"] == "non-reference" and abs(r["start"] - t["pos"]) <= 20 and r["strand"] == t["strand"] and r["family"] == t["family"] for r in by.get(l["name"], [])): s.add(l["name"])
    return s, by
sa, bya = ok(a); sb, byb = ok(b)
print("recovered off/on:", len(sa), len(sb), "lost:", sorted(sa - sb), "gained:", sorted(sb - sa))
ann_a = {r[0]: r for r in a["annotation"]}; ann_b = {r[0]: r for r in b["annotation"]}
for n in sorted(sa - sb)[:6]:
    print(n, "annotation off:", [r for r in a["annotation"] if r[0] == n], "on:", [r for r in b["annotation"] if r[0] == n])
    print("   liftover off:", [(r["type"], r["start"], r["family"], r.get("comment")) for r in bya.get(n, [])], "on:", [(r["type"], r["start"], r["family"], r.get("comment")) for r in byb.get(n, [])])
