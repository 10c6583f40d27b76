"""Debug helper (test infrastructure): where do the sorted anchor keys of the HIP engine and of the oracle differ on the bundled fixture?"""
import sys; sys.path.insert(0, '.')
import numpy as np
from telr_amd.aligner import Engine
from telr_amd.presets import preset
from telr_amd.fasta import read_fasta
from oracle import binding as ob
_, ts = read_fasta("tests/data/ref_38kb.fasta"); _, qs = read_fasta("tests/data/reads.fasta")
io, mo = preset("map-ont")
eng = Engine(0); gix = eng.index(ts, io); oix = ob.OracleIndex(ts, io)
o = oix.map(qs, mo, debug=True); res = gix.map(qs, mo); d = gix.debug_last_batch(len(qs))
off = o["anchor_off"]
for q in range(len(qs)):
    a = d["skeys"][off[q]:off[q+1]]; b = o["anchors"][off[q]:off[q+1]]
    same_set = np.array_equal(np.sort(a), np.sort(b))
    srt = np.all(a[:-1] <= a[1:]) if len(a) > 1 else True
    nbad = int((a != b).sum())
    if nbad:
        i = int(np.nonzero(a != b)[0][0])
        print(q, len(a), "bad", nbad, "same_set", same_set, "gpu_sorted", srt, "first", i, [hex(int(x)) for x in a[i:i+3]], [hex(int(x)) for x in b[i:i+3]])
