#!/usr/bin/env python3
"""Per-launch HBM traffic and VALU issue figures of the dominant kernel from the PMC summaries of a round
(tools/pmc_summary.py output) -> profiles/<tag>_pmc_k_dp_pk.json, which bench.py attaches to its `roofline` object when the
run is the same workload.

usage: pmc_to_json.py <tag> <config> <first-pass launches of the PMC run> [kernel]
  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies wide (>= 128 B per lane) reads at 1/2 (MI355X_MICROARCH.md,
  HBM section; tools/ubench/tb_pattern.hip calibrates it on this kernel's patterns), so it is doubled: an upper bound.
  VALU issue: SQ_INSTS_VALU wave-instructions / 1024 SIMDs against GRBM_GUI_ACTIVE / 8 XCDs cycles, at the MEASURED
  issue interval of the kernel's instruction mix (profiles/<tag>_valu_rate.txt: ~4.1 cycles per wave64 instruction for
  v_pk_*_i16 / v_bfi / v_alignbit / DPP, 2.4 for v_add / v_and / v_or / v_xor / v_mov)."""
import json, os, sys
tag, config, launches = sys.argv[1], sys.argv[2], int(sys.argv[3])
kern = sys.argv[4] if len(sys.argv) > 4 else "k_dp_pk"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def total(path, counter):
    for line in open(os.path.join(ROOT, "profiles", path)):
        f = line.split()
        if line.startswith(kern + " ") and counter in f:
            return float(f[-2])            # sum over the run's dispatches
    return None


fs, ws = total("%s_pmc_FETCH_SIZE.txt" % tag, "FETCH_SIZE"), total("%s_pmc_WRITE_SIZE.txt" % tag, "WRITE_SIZE")
vi, ga = total("%s_pmc_SQ.txt" % tag, "SQ_INSTS_VALU"), total("%s_pmc_SQ.txt" % tag, "GRBM_GUI_ACTIVE")
out = {"config": config, "kernel": kern, "first_pass_launches_in_pmc_run": launches,
       "traffic_bytes_per_launch": (2.0 * fs + ws) * 1024.0 / launches,
       "fetch_kb_per_launch_raw": fs / launches, "write_kb_per_launch": ws / launches,
       "source": "profiles/%s_pmc_{FETCH,WRITE}_SIZE.txt (KB, summed over the run's %s dispatches, / %d first-pass launches; FETCH doubled as for wide streaming reads: an upper bound)" % (tag, kern, launches)}
if vi and ga:
    cyc = ga / 8.0
    per_simd = vi / 1024.0
    out["valu_issue"] = {"wave_instructions_per_launch": vi / launches, "gpu_cycles_per_launch": cyc / launches,
                         "cycles_per_wave_instruction_per_simd": cyc / per_simd,
                         "measured_issue_interval_cycles": 4.1,
                         "valu_issue_frac": 4.1 * per_simd / cyc,
                         "note": "4.1 = measured issue interval of v_pk_*_i16 / v_bfi_b32 / v_alignbit_b32 / DPP moves at >= 2 waves per SIMD "
                                 "(profiles/%s_valu_rate.txt; v_fma_f32 control 2.2-2.6); the kernel's mix is ~90 %% such instructions" % tag}
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_pmc_k_dp_pk.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1))
