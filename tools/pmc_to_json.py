#!/usr/bin/env python3
"""Per-launch HBM traffic and VALU issue figures of the dominant kernel from the PMC summaries of a round
(tools/pmc_summary.py output) -> profiles/<tag>_pmc_k_dp_pk.json, which bench.py attaches to its `roofline` object (under
`from_profile`, with this file's provenance) when the run is the same workload.

usage: pmc_to_json.py <tag> <config> <algorithmic bytes per launch> [kernel]
  The number of first-pass launches is DERIVED from the dispatch counts of the summaries: every range runs the kernel twice
  (first pass + the small retry pass), so launches = dispatches / 2; the three counter passes must agree.
  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies wide (>= 128 B per lane) reads at 1/2 (MI355X_MICROARCH.md,
  HBM section; tools/ubench/tb_pattern.hip calibrates it on this kernel's patterns), so it is doubled: an upper bound.
  VALU issue: SQ_INSTS_VALU wave-instructions / 1024 SIMDs against GRBM_GUI_ACTIVE / 8 XCDs cycles, at the MEASURED
  issue interval of the kernel's instruction mix (profiles/<tag>_valu_rate.txt; the interval is read from that file: the
  median of the v_pk_*_i16 rows at 8 waves per SIMD, 4.1 if the file is absent)."""
import datetime, json, os, re, subprocess, sys
tag, config, alg_bytes = sys.argv[1], sys.argv[2], float(sys.argv[3])
kern = sys.argv[4] if len(sys.argv) > 4 else "k_dp_pk"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def row(path, counter):
    """(dispatches, sum over the run's dispatches)"""
    for line in open(os.path.join(ROOT, "profiles", path)):
        f = line.split()
        if line.startswith(kern + " ") and counter in f:
            return int(f[-3]), float(f[-2])
    raise SystemExit("no %s row for %s in %s" % (counter, kern, path))


def issue_interval():
    """the issue interval of the packed-int16 ops with the SIMD full (8 resident waves): the hardware's rate, which the
    kernel (2 waves per SIMD by its register budget) is priced against"""
    p = os.path.join(ROOT, "profiles", "%s_valu_rate.txt" % tag)
    vals = []
    if os.path.exists(p):
        for line in open(p):
            m = re.match(r"^v_pk_\S+_i16\s+W=8 .*=>\s+([0-9.]+) cycles", line)
            if m:
                vals.append(float(m.group(1)))
    vals.sort()
    return (vals[len(vals) // 2], "median of the v_pk_*_i16 rows of profiles/%s_valu_rate.txt at 8 waves per SIMD" % tag) if vals else (4.1, "round-2 measurement (profiles/r02_valu_rate.txt)")


(nf, fs), (nw, ws) = row("%s_pmc_FETCH_SIZE.txt" % tag, "FETCH_SIZE"), row("%s_pmc_WRITE_SIZE.txt" % tag, "WRITE_SIZE")
(nv, vi), (ng, ga) = row("%s_pmc_SQ.txt" % tag, "SQ_INSTS_VALU"), row("%s_pmc_SQ.txt" % tag, "GRBM_GUI_ACTIVE")
assert nf == nw == nv == ng and nf % 2 == 0, ("dispatch counts of the counter passes differ", nf, nw, nv, ng)
launches = nf // 2
interval, interval_src = issue_interval()
traffic = (2.0 * fs + ws) * 1024.0 / launches
try:
    sha = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()
except Exception:
    sha = None
out = {"config": config, "kernel": kern, "first_pass_launches_in_pmc_run": launches, "dispatches_in_pmc_run": nf,
       "head_sha": sha, "date": datetime.date.today().isoformat(),
       "command": "rocprofv3 --pmc <counter> -- python3 bench.py --config %s --data-cache ... --require-cache --no-cpu-baseline --loci 0 --no-stream-leg --steps 1 --warmup 0" % config,
       "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg_bytes, "traffic_over_algorithmic": traffic / alg_bytes if alg_bytes else None,
       "fetch_kb_per_launch_raw": fs / launches, "write_kb_per_launch": ws / launches,
       "source": "profiles/%s_pmc_{FETCH,WRITE}_SIZE.txt (KB, summed over the run's %d %s dispatches = %d first-pass launches + as many retry-pass launches; FETCH doubled as for wide streaming reads: an upper bound)" % (tag, nf, kern, launches)}
cyc = ga / 8.0
per_simd = vi / 1024.0
out["valu_issue"] = {"wave_instructions_per_launch": vi / launches, "gpu_cycles_per_launch": cyc / launches,
                     "cycles_per_wave_instruction_per_simd": cyc / per_simd,
                     "measured_issue_interval_cycles": interval, "issue_interval_source": interval_src,
                     "valu_issue_frac": interval * per_simd / cyc,
                     "note": "issue interval of v_pk_*_i16 / v_bfi_b32 / v_alignbit_b32 / DPP moves at >= 2 waves per SIMD (v_fma_f32 control 2.2-2.6); the kernel's mix is ~90 % such instructions"}
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_pmc_k_dp_pk.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1))
