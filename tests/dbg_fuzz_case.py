"""debug aid: python tests/dbg_fuzz_case.py <seed> -- one fuzz configuration, the chain rows that differ between engine and oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["TELR_DEBUG"] = "1"
import numpy as np
import fuzz_parity
from telr_amd.aligner import Engine
from oracle import binding as ob
seed = int(sys.argv[1])
pname, io, mo, genome, reads, qtarget, edge = fuzz_parity.draw_case(seed)
S = lambda a: bytes(a).decode() if not isinstance(a, str) else a
eng = Engine(0)
T = [S(g) for g in genome]; Q = [S(r) for r in reads]
oref = ob.OracleIndex(T, io).map(Q, mo, qtarget=qtarget, debug=True)
gix = eng.index(T, io); res = gix.map(Q, mo, qtarget=qtarget)
dbg = gix.debug_last_batch(len(Q))
a, b = dbg["chains"], oref["chains"]
print("chains", a.shape, b.shape, "target lengths", [len(t) for t in T])
for i in range(min(len(a), len(b))):
    if (a[i] != b[i]).any():
        print("row", i, "engine", a[i].tolist(), "oracle", b[i].tolist(), "qlen", len(Q[a[i][0]]), "qtarget", None if qtarget is None else int(qtarget[a[i][0]]))
