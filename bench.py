#!/usr/bin/env python3
"""bench.py — stage-1 long-read alignment throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (sketch -> seed -> sort -> chain -> back-track ->
banded DP + trace-back -> records/CIGARs on the host) over the whole synthetic read set
of BASELINE.json configs[1] (chr2L-sized genome, 10k ONT-like reads ~20x, 200 spiked TE
insertions).  The reference index and the packed reads are resident in HBM before the
timed region.  One process per GPU; reads are sharded per rank (each rank maps its own
read set against a replicated index, weak scaling, no data-path collective).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The engine keeps up to eight streams busy (tail classes of the DP, back-tracking tiers, copies).  The HIP runtime maps
# streams onto 4 hardware queues by default and RCCL's own streams take some of them: measured under torch.distributed.run,
# 35.8 ms per step with 4 queues against 29.4 ms with 8 (no difference without RCCL in the process).  Must be set before
# the first HIP call of the process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-len", type=int, default=23513712)
    ap.add_argument("--reads", type=int, default=10000)
    ap.add_argument("--read-bases", type=int, default=470_000_000)
    ap.add_argument("--insertions", type=int, default=200)
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="0 = auto (about 15-25 s of CPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preset", default="map-ont")
    ap.add_argument("--fill-band-q4", type=int, default=0, help="experiment: override the preset's first-pass band factor (0 = preset)")
    ap.add_argument("--loci", type=int, default=200, help="candidate loci for the TE-loci/s leg (0 = skip)")
    return ap.parse_args()


def usable_cpus():
    """CPUs this process may really use: affinity mask, narrowed by a cgroup v2 quota when there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except Exception:
        pass
    return n


def cpu_baseline(ref_str, reads, io, mo, n_sample, gbp_of):
    """The CPU oracle ("port") timed on a bounded sample of the same read set."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as ob
    cores = usable_cpus()
    buf, off, ln = reads
    t0 = time.time()
    oix = ob.OracleIndex([ref_str], io)
    t_index = time.time() - t0
    n_sample = min(n_sample, len(ln))
    seqs = [bytes(buf[off[i]:off[i] + ln[i]]).decode() for i in range(n_sample)]
    shards = [seqs[i::cores] for i in range(cores)]
    shards = [s for s in shards if s]

    def work(s):
        r = oix.map(s, mo)
        prim = r["alns"][(r["alns"]["flags"] & 1) != 0]
        return int(prim["qlen"].sum())
    t0 = time.time()
    with ThreadPoolExecutor(max_workers=len(shards)) as ex:
        aligned = sum(ex.map(work, shards))
    dt = time.time() - t0
    return {"value": aligned / dt / 1e9, "unit": "Gbp/s", "cores": len(shards), "kind": "port",
            "sample": "first %d reads (%d bases) of the same read set, oracle/telr_oracle.c, %d threads, %.1f s; index build %.1f s excluded"
                      % (n_sample, sum(len(s) for s in seqs), len(shards), dt, t_index)}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # ranks of one node share its CPUs: each engine sizes its host pool for its share (the library's default is 1.5 x the
    # CPUs the process may use, which every rank would claim for itself)
    lws = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if lws > 1 and "TELR_HOST_THREADS" not in os.environ:
        os.environ["TELR_HOST_THREADS"] = str(max(4, min(48, usable_cpus() * 3 // (2 * lws))))
    import torch
    dist = None
    if world > 1 or "RANK" in os.environ:      # launched by torch.distributed.run (also at world size 1)
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from telr_amd.aligner import Engine
    from telr_amd.presets import preset
    from telr_amd import synth

    io, mo = preset(a.preset)
    if a.fill_band_q4:
        mo.fill_band_q4 = a.fill_band_q4
    t0 = time.time()
    # every rank: same genome/insertions (seed), its own reads (seed + rank)
    d = synth.make_stage1_dataset(seed=20261002, genome_len=a.genome_len, n_reads=a.reads, total_bases=a.read_bases,
                                  n_ins=a.insertions, read_seed=20261002 + 1000 * (rank + 1))
    t_gen = time.time() - t0
    ref_str = bytes(d["ref"]).decode()
    eng = Engine(local)
    t0 = time.time()
    ix = eng.index([ref_str], io)
    t_index = time.time() - t0
    qs = eng.seqset(d["reads"])
    n_bases = qs.bases()

    # Steps are streamed the way a stage-1 run over many read batches would be: telr_map returns when the alignment
    # records are complete, the DMA of the CIGAR array finishes in the background, and a result is released only after
    # the next call has been issued.  Everything, the last DMA included, is complete before the closing synchronisation.
    held = []

    def step():
        r = ix.map_raw(qs, mo)
        L = eng.L
        n = L.telr_result_count(r)
        from telr_amd.aligner import _np_from
        from telr_amd._abi import ALN_DTYPE
        al = _np_from(L.telr_result_alns(r), n, ALN_DTYPE)
        while held:
            ix.free_raw(held.pop())
        held.append(r)
        prim = al[(al["flags"] & 1) != 0]
        return int(prim["qlen"].sum()), al

    def sync():
        torch.cuda.synchronize(local)
        if dist is not None:
            dist.barrier()

    step(); step()      # set-up, not warm-up steps: the first calls size the engine's grow-only scratch and pin the TWO
                        # result buffers that are alive at any time when steps are streamed
    for _ in range(a.warmup):
        step()
    stage_tot = {}
    sync()
    t0 = time.time()
    aligned = 0
    for _ in range(a.steps):
        b, al = step()
        aligned += b
        for k, v in eng.stage_ms().items():
            stage_tot[k] = stage_tot.get(k, 0.0) + v
    if held:
        eng.L.telr_result_wait(held[-1])      # the last step's CIGAR array is home as well
    sync()
    dt = time.time() - t0
    while held:
        ix.free_raw(held.pop())
    ctr = eng.counters()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
        s = torch.tensor([aligned], dtype=torch.float64, device="cuda"); dist.all_reduce(s, op=dist.ReduceOp.SUM); aligned = float(s.item())
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    value = aligned / dt / 1e9
    # truth check on the last step: primary alignment overlaps the simulated origin (haplotype coordinates differ from
    # reference coordinates by at most the inserted TE bases upstream, so compare loosely)
    prim = al[(al["flags"] & 1) != 0]
    frac_mapped = len(np.unique(prim["qid"])) / max(1, len(d["reads"][2]))
    # roofline of the dominant kernel: k_dp_pk, the packed-int16 gap-fill DP (DP classes 10-17, gap fills by band width, in one cost-ordered launch).  Its
    # launch duration is measured live with HIP events on the engine's own stream (telr_stage_ms: "k_dp_pk"); its
    # algorithmic bytes are counted per problem by the library: 2-bit query+target bases read once, 4 B per CIGAR
    # run, 32 B result.
    cls = eng.dp_classes()
    PK = list(range(10, 18))
    k_name = "k_dp_pk"
    k_ms = stage_tot.get("k_dp_pk", 0.0) / a.steps
    k_bytes = float(cls[PK, 3].sum())
    achieved = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    def _pmc2(path, kern, counter):
        for line in open(os.path.join(ROOT, "profiles", path)):
            f = line.split()
            if line.startswith(kern) and counter in f:
                return float(f[-2]) / 3.0
        return None
    traffic, traffic_src = None, None
    try:   # HBM bytes per launch from the committed PMC passes of this same command (profiles/, separate --pmc runs)
        def _pmc(path, kern):
            for line in open(os.path.join(ROOT, "profiles", path)):
                if line.startswith(kern):
                    return float(line.split()[-2]) / 3.0     # sum over the PMC run's three map calls (two set-up calls + one step), per call
            return None
        fs = _pmc("r01_pmc_FETCH_SIZE.txt", k_name + " "); ws = _pmc("r01_pmc_WRITE_SIZE.txt", k_name + " ")
        if fs is not None and ws is not None and a.reads == 10000 and a.read_bases == 470_000_000:
            traffic = (2.0 * fs + ws) * 1024.0      # gfx950: FETCH_SIZE counts wide reads at 1/2 (MI355X_MICROARCH.md, HBM)
            traffic_src = ("profiles/r01_pmc_{FETCH,WRITE}_SIZE.txt (KB per map call; FETCH doubled as for wide streaming reads: an upper bound, "
                           "scattered 64-byte reads calibrate at 1.0 and 40-byte write pieces at 1.47, tools/ubench/tb_pattern.hip)")
    except Exception:
        pass
    issue_frac = None
    try:   # integer-issue utilisation of the same kernel from the committed SQ counter pass (the roof that actually binds it)
        vi = _pmc2("r01_pmc_SQ.txt", k_name + " ", "SQ_INSTS_VALU"); ga = _pmc2("r01_pmc_SQ.txt", k_name + " ", "GRBM_GUI_ACTIVE")
        if vi and ga:
            issue_frac = (vi * 4.0 / 1024.0) / (ga / 8.0)      # 4 cycles per wave64 VALU instruction, 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs
    except Exception:
        pass
    dp_ms = stage_tot.get("dp", 0.0) / a.steps
    # whole-path algorithmic bytes (SURVEY 8d formula) for reference
    path_bytes = (ctr["query_bases"] / 4.0 + 32.0 * ctr["minimizers"] + 16.0 * ctr["probes"] + 48.0 * ctr["anchors"]
                  + ctr["window_bases"] / 4.0 + 4.0 * ctr["cigar_ops"] + 64.0 * ctr["records"])
    gpu_ms = sum(v for k, v in stage_tot.items() if k not in ("select_host", "assemble_host", "index_build", "k_dp_reg", "map_wall", "k_traceback", "k_dp_pk")) / a.steps
    out = {
        "metric": "gbp_aligned_per_s", "value": value, "unit": "Gbp/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: synthetic chr2L-size genome (%d bp) + %d ONT-like reads (%.0f Mbp per GPU, 10%% error) + %d spiked TE insertions, preset %s, stage-1 reads->reference"
                               % (a.genome_len, a.reads, n_bases / 1e6, a.insertions, a.preset),
                   "reads_per_gpu": a.reads, "read_bases_per_gpu": n_bases, "parallelism": "reads sharded x%d, index replicated" % world,
                   "streaming": "telr_map returns with the records; the CIGAR DMA of step k overlaps step k+1 (all complete inside the timed region)"},
        "roofline": {"bound": "hbm", "kernel": k_name, "dp_classes": PK, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                     "traffic": traffic, "traffic_source": traffic_src, "valu_issue_frac": issue_frac, "launch_ms": k_ms, "algorithmic_bytes_per_launch": k_bytes,
                     "problems_per_launch": int(cls[PK, 0].sum()), "cells_per_launch": int(cls[PK, 1].sum()),
                     "gcups": float(cls[PK, 1].sum()) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None,
                     "note": "integer DP is VALU-issue bound, not HBM bound (DESIGN.md, Rooflines); all DP kernels together: %.1f ms, %.0f GCUPS"
                             % (dp_ms, ctr["dp_cells"] / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0)},
        "stage_ms_per_step": {k: v / a.steps for k, v in stage_tot.items()},
        "dp_classes": {str(c): [int(x) for x in cls[c]] for c in range(cls.shape[0]) if cls[c, 0]},
        "path_algorithmic_GBps": path_bytes / (gpu_ms * 1e-3) / 1e9 if gpu_ms > 0 else None,
        "frac_reads_mapped": frac_mapped, "index_build_s": t_index, "datagen_s": t_gen, "device": eng.device_name(),
        "counters": ctr, "dp_retries": int(eng.L.telr_debug_dp_retries(eng.h)),
    }
    if a.loci > 0:
        # second half of the BASELINE metric: TE loci/s through the per-locus bundle (S4, S5, S6 fw+rc + depth + AF,
        # S7 x2 + liftover).  Asm10 index of the same reference; loci = the spiked insertions (truth-derived contigs).
        from telr_amd import locus_pipeline
        loci = synth.make_loci_from_dataset(d, min(a.loci, len(d["insertions"])))
        io10, _ = preset("asm10")
        ix10 = eng.index([ref_str], io10)
        lib_names = ["fam%d" % i for i in range(len(d["library"]))]
        lib = [bytes(x).decode() for x in d["library"]]
        # window reads are taken from the stage-1 read set already on the device (telr_seqset_subset)
        locus_pipeline.run_loci(eng, ix10, ["chr2L"], lambda ch: ref_str, loci, lib_names, lib, read_set=qs)      # warm-up (sizes the scratch)
        t0 = time.time()
        lres = locus_pipeline.run_loci(eng, ix10, ["chr2L"], lambda ch: ref_str, loci, lib_names, lib, read_set=qs)
        t_loci = time.time() - t0
        good = 0
        by = {"_".join(r["ID"].split("_")[:3]): r["report"] for r in lres["liftover"]}
        for l in loci:
            r = by.get(l["name"])
            if r and r["type"] == "non-reference" and abs(r["start"] - l["truth"]["pos"]) <= 20 and r["strand"] == l["truth"]["strand"] and r["family"] == l["truth"]["family"]:
                good += 1
        out["te_loci_per_s"] = len(loci) / t_loci
        out["te_loci"] = {"n": len(loci), "seconds": t_loci, "recovered_exact_family_strand_pos20": good,
                          "note": "host glue (Python) included; per-locus inputs are truth-derived (no Sniffles/wtdbg2 on the box)"}
    if not a.no_cpu_baseline:
        ns = a.cpu_sample_reads or max(8, int(3.0e7 * usable_cpus() / max(1.0, n_bases / len(d["reads"][2]))))   # ~15-20 s of CPU work
        out["cpu_baseline"] = cpu_baseline(ref_str, d["reads"], io, mo, ns, None)
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"] if out["cpu_baseline"]["value"] > 0 else None
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
