"""Which DP classes does the large-indel fuzz (tests/fuzz_parity.py, sv=True) reach?  Runs 40 big cases on the `ngmlr-*` presets with
the stage-by-stage comparison against the oracle and sums `telr_last_dp_classes` (through gpurun: python tools/fuzz_class_probe.py)."""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import fuzz_parity
from telr_amd.aligner import Engine
eng = Engine(0)
tot = None
for it in range(40):
    pname, io, mo, genome, reads, qtarget, er = fuzz_parity.draw_case(78 * 1000 + it, True, True, ["ngmlr-ont", "ngmlr-pacbio"])
    from test_gpu_parity import compare_all
    compare_all(eng, genome, reads, io, mo, qtarget=qtarget)
    c = eng.dp_classes()
    tot = c if tot is None else tot + c
for k in range(tot.shape[0]):
    if tot[k, 0]: print("class", k, "problems", int(tot[k, 0]), "cells", int(tot[k, 1]))
