"""debug aid: python tests/dbg_fuzz_records.py <seed> -- one fuzz configuration, the records that differ between engine and oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzz_parity
from telr_amd.aligner import Engine
from oracle import binding as ob
seed = int(sys.argv[1])
pname, io, mo, genome, reads, qtarget, edge = fuzz_parity.draw_case(seed)
if len(sys.argv) > 2:
    mo.ext_band = int(sys.argv[2])
S = lambda a: bytes(a).decode() if not isinstance(a, str) else a
eng = Engine(0)
T = [S(g) for g in genome]; Q = [S(r) for r in reads]
o = ob.OracleIndex(T, io).map(Q, mo, qtarget=qtarget)
r = eng.index(T, io).map(Q, mo, qtarget=qtarget)
print(pname, "ext_band", mo.ext_band, "ext_max", mo.ext_max, "zdrop", mo.zdrop, "records", len(r.alns), len(o["alns"]), "cx", mo.cx_scale)
F = ("qid", "tid", "qs", "qe", "ts", "te", "mlen", "blen", "dp_score", "flags", "n_cigar")
for i in range(min(len(r.alns), len(o["alns"]))):
    a, b = r.alns[i], o["alns"][i]
    if any(a[f] != b[f] for f in F):
        print("rec", i, "qlen", a["qlen"], "\n  engine", [int(a[f]) for f in F], "\n  oracle", [int(b[f]) for f in F])
print("dp classes", {c: [int(x) for x in row] for c, row in enumerate(eng.dp_classes()) if row[0]})
