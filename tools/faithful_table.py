#!/usr/bin/env python3
"""The drift table of the oracle's experiment bits (oracle/telr_oracle.c: MFX_*) for DESIGN.md section 2: every omission of the
engine's spec against minimap2 2.22's published behaviour, on every gate workload.  CPU only; ~10 minutes on 8 cores.
usage: python tools/faithful_table.py [--hard] [reads per workload, default 300] [workload, ...]
--hard: every workload on the HARD genome (telr_amd/synth.py: tandem arrays, microsatellites, low-complexity stretches, segmental
duplications, satellite blocks next to the insertions; reads with error bursts) -- profiles/r06_faithful_table_hard.md"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_faithful_gate as g

hard = "--hard" in sys.argv
if hard:
    sys.argv.remove("--hard")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
print("| workload | omission | records | records changed | coordinates changed | DP score changed | CIGAR changed (same coordinates) |")
print("|---|---|---|---|---|---|---|")
only = sys.argv[2:]
for kind, k in (("flanks-asm10", 2 * n), ("clr-map-pb", n), ("clr-ngmlr-pacbio", n), ("ont-ngmlr-ont", n), ("ont-map-ont", n), ("c4-density", n)):
    if only and kind not in only:
        continue
    for name, r in g.bit_table(kind, k, hard=hard):
        print("| %s%s | %s | %d | %.2f %% | %.2f %% | %.2f %% | %s |" % (kind, " (hard genome)" if hard else "", name, r["n"], 100 * r["core"], 100 * r["coord"], 100 * r["score"], ("%.2f %%" % (100 * r["cigar"])) if "cigar" in r else "-"), flush=True)
