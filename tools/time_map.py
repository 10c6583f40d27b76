import sys, time; sys.path.insert(0, '.')
import numpy as np
from telr_amd.aligner import Engine, _np_from
from telr_amd._abi import ALN_DTYPE
from telr_amd.presets import preset
from telr_amd import synth
d = synth.make_stage1_dataset()
io, mo = preset("map-ont"); eng = Engine(0); ref = bytes(d["ref"]).decode(); ix = eng.index([ref], io); qs = eng.seqset(d["reads"])
for it in range(3):
    t0 = time.time(); r = ix.map_raw(qs, mo); t1 = time.time()
    n = eng.L.telr_result_count(r); al = _np_from(eng.L.telr_result_alns(r), n, ALN_DTYPE); t2 = time.time()
    ix.free_raw(r); t3 = time.time()
    st = eng.stage_ms()
    acc = sum(v for k, v in st.items() if k not in ("map_wall", "k_dp_reg", "k_traceback", "k_dp_pk"))
    print("map_raw %.1f ms (lib wall %.1f, sum of stages %.1f) | copy alns %.1f | free %.1f" % ((t1-t0)*1e3, st["map_wall"], acc, (t2-t1)*1e3, (t3-t2)*1e3))
