#!/usr/bin/env python3
"""bench.py — the BASELINE.json metric: long-read Gbp aligned/s + TE loci/s on MI355X.

Default workload = BASELINE configs[2] (the configuration the metric is quoted on): a synthetic genome with the eight
dm6 arm lengths (137,567,484 bp, 15 % diverged TE copies of a 127-family library), 30x ONT-like reads (~4.1 Gbp, mean
9 kb, 10 % errors 4:2:4), 1,000 spiked TE insertions at allele frequencies 0.25 / 0.5 / 1.0, preset `map-ont`.
`--config c1` = configs[1] (chr2L-size genome, 10,000 reads, 200 insertions), `c3` = configs[3] (same genome, CLR-like
reads 13 % 1:5:4, the NGMLR-style convex-gap preset `ngmlr-pacbio`), `c4` = configs[4] (chr22-size target with an 11-Mb
leading N block, 50x, 1,300-family library, 3,000 insertions).

A "step" = one pass of the stage-1 hot path (sketch -> seed -> sort -> chain -> back-track -> banded DP + trace-back ->
records/CIGARs on the host) over the rank's WHOLE read set: up to 1.6 Gbp is one range, a larger set streams through in
an EVEN number of equal ranges of at most 1.4 Gbp (configs[2]: four of 0.96), two in flight (each also bounded by an anchor budget at the density the index has shown).
Index and packed reads are resident in HBM before the timed region (`value`); `value_incl_h2d` adds the packing +
upload of the reads.  The second half of the metric, TE loci/s, runs the per-locus bundle (S4, S5, S6 fw+rc + depth +
AF, S7 x2 + liftover) on window reads selected from the ENGINE'S OWN stage-1 records (TELR_assembly.py:384-415).

Multi-GPU: `--gpus N` without RANK in the environment starts N ranks itself (python -m torch.distributed.run, before
anything touches the GPU) and relays rank 0's line; under torch.distributed.run it is one of the ranks.  `--scaling
strong` (default): ONE fixed read set dealt to the ranks in blocks by cumulative bases (shard.shard_reads), index
replicated (built by every rank), no collective on the stage-1 data path; the loci are dealt by LPT, their window reads
travel to the owner in one all-to-all and the per-locus table is merged by ONE all-gather (RCCL).  `--scaling weak`:
every rank maps its own 30x read set.

Next to `value`, `roofline` and `cpu_baseline` the line carries: `stage1_to_sorted_bam` (reads resident -> telr_map -> coordinate-
sorted BAM + .bai built on the device, the reference's real stage-1 hand-off; at N > 1 also `job_bam`: the ONE file of the job,
written by rank 0 from the gathered records), `stage1_from_files` (--files-leg: FASTA files -> telr_alignment.alignment()),
`value_streaming_incl_h2d`, `te_loci` (+ `polish_pileup`), and at N = 1 `expected_strong_scaling` (rank 0's shard of a 2 / 4 / 8-rank
run mapped alone on this GPU).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
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

CONFIGS = {
    # name: (BASELINE.json config, chromosomes, coverage, read errors (sub, ins, del), preset, insertions, families, lead N)
    "c1": dict(label="configs[1]", preset="map-ont", err=(0.04, 0.02, 0.04)),
    "c2": dict(label="configs[2]", genome="dm6", coverage=30.0, preset="map-ont", err=(0.04, 0.02, 0.04), n_ins=1000, n_fam=127, lead_n=0),
    # configs[2] on the HARD genome (round 6; telr_amd/synth.py HARD: tandem arrays, microsatellites, low-complexity stretches, segmental duplications,
    # a satellite block next to 15 % of the insertions; reads with error bursts) -- not a BASELINE configuration: the same workload where the
    # aligner heuristics this engine leaves out (seed rescue, max_chain_skip, RMQ chaining) and the over-size sort path would bite
    "c2r": dict(label="configs[2] on the hard genome", genome="dm6", coverage=30.0, preset="map-ont", err=(0.04, 0.02, 0.04), n_ins=1000, n_fam=127, lead_n=0, hard=True),
    "c3": dict(label="configs[3]", genome="dm6", coverage=30.0, preset="ngmlr-pacbio", err=(0.013, 0.065, 0.052), n_ins=1000, n_fam=127, lead_n=0),
    "c4": dict(label="configs[4]", genome="chr22", coverage=50.0, preset="map-ont", err=(0.04, 0.02, 0.04), n_ins=3000, n_fam=1300, lead_n=11_000_000),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--coverage", type=float, default=0.0, help="override the configuration's coverage (experiments)")
    ap.add_argument("--genome-scale", type=float, default=1.0, help="scale every chromosome length (experiments / CPU smoke tests)")
    ap.add_argument("--genome-len", type=int, default=23513712, help="c1 only")
    ap.add_argument("--reads", type=int, default=10000, help="c1 only")
    ap.add_argument("--read-bases", type=int, default=470_000_000, help="c1 only")
    ap.add_argument("--insertions", type=int, default=0, help="override the number of spiked insertions")
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="0 = auto (about 15-25 s of CPU work)")
    ap.add_argument("--cpu-sample-seed", type=int, default=20261002, help="seed of the random read sample the CPU oracle maps (cpu_baseline, parity)")
    ap.add_argument("--require-cache", action="store_true", help="with --data-cache: fail instead of generating (profiled runs must not fork the generator)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-default-aligner-leg", action="store_true", help="skip value_reference_default_aligner (the same timed steps with the reference's default aligner, ngmlr -x ont|pacbio)")
    ap.add_argument("--default-aligner-parity-reads", type=int, default=12000, help="reads of the oracle parity sample of that leg (0 = none)")
    ap.add_argument("--no-upstream-check", action="store_true", help="skip reference_cpu_path (the upstream tools on PATH, tools/crosscheck_upstream.py)")
    ap.add_argument("--upstream-sample-reads", type=int, default=2000, help="reads of the sample the upstream tools map when they are on PATH")
    ap.add_argument("--preset", default="", help="override the configuration's preset")
    ap.add_argument("--fill-band-q4", type=int, default=0, help="experiment: override the preset's first-pass band factor (0 = preset)")
    ap.add_argument("--map-opt", default="", help="experiment: map-option fields laid over the preset of the timed map and of the CPU oracle, e.g. cx_scale=20,cx_open=100 (the loci leg keeps its presets)")
    ap.add_argument("--loci", type=int, default=-1, help="candidate loci for the TE-loci/s leg (-1 = all spiked insertions, 0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for the CPU smoke test of the launcher)")
    ap.add_argument("--data-cache", default="", help="directory: the rank's generated data set is stored there / loaded from there (profiling runs: no forked generator)")
    ap.add_argument("--one-gpu", action="store_true", help="smoke test of the N>1 code path on a 1-GPU box: every rank uses device 0 (use with --backend gloo; RCCL refuses two ranks on one device)")
    ap.add_argument("--no-bam-twin", action="store_true", help="BAM leg without TELR_MF_KEEP_CIGARS: the writer uploads the CIGAR array again")
    ap.add_argument("--bam-sha", action="store_true", help="report the SHA-256 of the BAM files the legs write (the N-rank job BAM equals the 1-rank BAM byte for byte)")
    ap.add_argument("--force-exchange", action="store_true", help="run the N>1 code path of the loci leg (window-read all-to-all, pooled read set, all-gather) at world size 1 too: under torch.distributed.run on a 1-GPU box this drives the collectives through RCCL on device tensors")
    ap.add_argument("--flank-parity", action="store_true", help="S7 at full size: the first and last 500 bases of every locus contig (2 x loci flanks) through `asm10 -N 10` against the full reference index, engine vs CPU oracle, record by record (one-off parity evidence; the oracle indexes the reference on one thread)")
    ap.add_argument("--no-shard-leg", action="store_true", help="skip expected_strong_scaling (the shard of rank 0 of a 2 / 4 / 8-rank run mapped alone on this GPU)")
    ap.add_argument("--no-stream-leg", action="store_true", help="skip the streaming host-inclusive measurement (a second context uploads the next read batch while the first maps)")
    ap.add_argument("--bam-leg", default="device", choices=["none", "host", "device"], help="stage 1 to the Sniffles hand-off (TELR_alignment.py:103-114): reads resident -> telr_map -> coordinate-sorted BAM + .bai under --bam-dir; host = the library's host-thread writer, device = record bodies / sort / BGZF on the GPU")
    ap.add_argument("--bam-dir", default="/dev/shm")
    ap.add_argument("--bam-level", type=int, default=1)
    ap.add_argument("--files-leg", action="store_true", help="stage 1 as the reference runs it, from FILES: the read set and the reference are written as FASTA under --bam-dir, then telr_alignment.alignment(bam, reads.fa, ref.fa, ...) is timed end to end (parse, pack, upload, index, map, sorted BAM + .bai)")
    ap.add_argument("--poa-parity", type=int, default=0, help="polish leg: the device consensus (window POA and pile-up) of the first N loci against the CPU oracle, contig for contig")
    ap.add_argument("--no-polish-leg", action="store_true", help="skip the (untimed-for-the-metric) device polishing pass over the loci")
    ap.add_argument("--no-bam-prepare", action="store_true", help="do not create / allocate / map the BAM file in the background while the reads are mapped")
    ap.add_argument("--dry-launch", action="store_true", help="launcher smoke test: ranks initialise torch.distributed, report and exit (no GPU work)")
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


def launch_ranks(a):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as children of THIS process, which has
    not touched the GPU (no torch / HIP import so far), relay their output and exit with their return code."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % a.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # only rank 0's JSON line belongs on stdout (the gloo backend, for one, prints its connection banner there)
    pr = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in pr.stdout:
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
    sys.stdout.flush()
    return pr.wait()


PMC_PROFILE = "r06_pmc_k_dp_pk.json"       # per-launch counters of the dominant kernel, collected by tools/collect_profiles.sh
PMC_PROFILE_NG = "r06_ngmlr_ont_c2_pmc_k_dp_pk.json"      # the same for the default-aligner leg (tools/collect_ngmlr_profiles.sh + tools/pmc_to_json.py)
PARITY_FIELDS = ("tid", "qlen", "qs", "qe", "tlen", "ts", "te", "mlen", "blen", "score", "subsc", "dp_score", "cnt", "n_sub", "parent", "n_cigar", "flags", "mapq")


def _read_digests(alns, cigars, qid_map=None):
    """{read: bytes of its records (the fields the parity tests compare) + their CIGAR words, in record order}"""
    import numpy as np
    out = {}
    qid = alns["qid"] if qid_map is None else qid_map[alns["qid"]]
    tab = np.stack([alns[f].astype(np.int64) for f in PARITY_FIELDS], axis=1) if len(alns) else np.zeros((0, len(PARITY_FIELDS)), np.int64)
    off, n = alns["cigar_off"], alns["n_cigar"]
    for k in range(len(alns)):
        q = int(qid[k])
        out[q] = out.get(q, b"") + tab[k].tobytes() + cigars[off[k]:off[k] + n[k]].tobytes()
    return out


def cpu_baseline(ref_strs, reads, io, mo, n_sample, gpu_index=None, seed=20261002):
    """The CPU oracle ("port") timed on a bounded sample of the same read set; with `gpu_index`, its records for the sample are
    also compared with the engine's, read by read (full-size parity evidence: same index, same reads, outside the timed leg)."""
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    from oracle import binding as ob
    cores = usable_cpus()
    buf, off, ln = reads
    t0 = time.time()
    oix = ob.OracleIndex(ref_strs, io)
    t_index = time.time() - t0
    n_sample = min(n_sample, len(ln))
    pick = np.sort(np.random.default_rng(seed).choice(len(ln), size=n_sample, replace=False))
    seqs = [bytes(buf[off[i]:off[i] + ln[i]]).decode() for i in pick]
    shards = [seqs[i::cores] for i in range(cores)]
    shards = [s for s in shards if s]
    parts = [None] * len(shards)

    def work(k):
        r = oix.map(shards[k], mo)
        parts[k] = (r["alns"], r["cigars"])
        prim = r["alns"][(r["alns"]["flags"] & 1) != 0]
        return int(prim["qlen"].sum())
    t0 = time.time()
    with ThreadPoolExecutor(max_workers=len(shards)) as ex:
        aligned = sum(ex.map(work, range(len(shards))))
    dt = time.time() - t0
    out = {"value": aligned / dt / 1e9, "unit": "Gbp/s", "cores": len(shards), "kind": "port",
           "sample": "%d reads drawn at random (seed %d; %d bases) from the same read set, mapped against the same full-size index, oracle/telr_oracle.c, %d threads, %.1f s; "
                     "index build %.1f s (one thread) excluded" % (n_sample, seed, sum(len(s) for s in seqs), len(shards), dt, t_index)}
    if gpu_index is not None:
        want = {}
        for k, (al, cg) in enumerate(parts):
            want.update(_read_digests(al, cg, qid_map=np.arange(len(shards[k]), dtype=np.int32) * len(shards) + k))
        res = gpu_index.map(seqs, mo)
        got = _read_digests(res.alns, res.cigars)
        bad = [q for q in set(want) | set(got) if want.get(q) != got.get(q)]
        out["parity"] = {"reads": n_sample, "records_oracle": int(sum(len(p[0]) for p in parts)), "records_engine": int(len(res.alns)),
                         "reads_differing": len(bad), "identical": not bad,
                         "what": "every record (all fields) and every CIGAR of the sampled reads, engine vs CPU oracle on the same full-size index"}
    return out


def build_dataset(a, cfg, rank, world, lws):
    """-> dict(names, ref [uint8 arrays], library, reads (buf, off, len), read_gid, insertions-derived loci source, workload text).
    Runs before the process touches the GPU (the read generator forks workers)."""
    import numpy as np
    from telr_amd import synth, shard
    procs = max(1, usable_cpus() // max(1, lws))
    if a.config == "c1":
        d = synth.make_stage1_dataset(seed=20261002, genome_len=a.genome_len, n_reads=a.reads, total_bases=a.read_bases,
                                      n_ins=a.insertions or 200, read_seed=20261002 + 1000 * ((rank if a.scaling == "weak" else 0) + 1))
        buf, off, ln = d["reads"]
        gid = np.arange(len(ln))
        if a.scaling == "strong" and world > 1:
            mine = np.array(shard.shard_reads(ln, world)[rank], np.int64)
            parts = [buf[off[i]:off[i] + ln[i]] for i in mine]
            ln = ln[mine]; buf = np.concatenate(parts); off = np.cumsum(ln.astype(np.int64)) - ln; gid = mine
        loci = synth.make_loci_from_dataset(d, len(d["insertions"]))
        for l in loci:
            l["chrom"] = "chr2L"; l["start"] = l["truth"]["pos"]; l["end"] = l["truth"]["pos"] + 1; l["truth"]["chrom"] = "chr2L"
            l.pop("reads", None); l.pop("read_idx", None)
        text = "synthetic chr2L-size genome (%d bp) + %d ONT-like reads (%.0f Mbp, 10%% error) + %d spiked TE insertions" % (
            a.genome_len, a.reads, float(d["reads"][2].sum()) / 1e6, len(d["insertions"]))
        return dict(names=["chr2L"], ref=[d["ref"]], library=d["library"], reads=(buf, off.astype(np.int64), ln.astype(np.int32)), read_gid=gid,
                    loci=loci, text=text, total_reads=len(d["reads"][2]), total_bases=int(d["reads"][2].sum()))
    chroms = synth.DM6_ARMS if cfg["genome"] == "dm6" else synth.CHR22
    if a.genome_scale != 1.0:
        chroms = [(n, max(30000, int(L * a.genome_scale))) for n, L in chroms]
    lead_n = int(cfg["lead_n"] * a.genome_scale)
    hard = synth.HARD if cfg.get("hard") else None
    g = synth.make_genome(20261002, chroms, n_fam=cfg["n_fam"], n_ins=a.insertions or cfg["n_ins"], lead_n=lead_n, threads=min(procs, 8), hard=hard)
    cov = a.coverage or cfg["coverage"]
    plan = synth.plan_reads(g, cov, read_seed=(20261002 + 1000 * (rank + 1)) if a.scaling == "weak" else None)
    if a.scaling == "strong" and world > 1:
        blocks = np.array(shard.shard_reads(synth.block_bases(plan), world)[rank], np.int64)
    else:
        blocks = np.arange(plan["n_blocks"])
    buf, off, ln, gid = synth.materialize_reads(g, plan, blocks, err=cfg["err"], procs=procs, burst=hard["burst"] if hard else None)
    loci = synth.make_loci(g)
    nb = int(sum(len(r) for r in g["ref"]))
    text = "synthetic %s-size genome (%d sequences, %d bp%s, %d-family TE library, 15%% TE-derived) + %.0fx %s-like reads (%d reads, %.2f Gbp planned, errors sub:ins:del %.3f:%.3f:%.3f) + %d spiked TE insertions (AF 0.25/0.5/1.0)" % (
        cfg["genome"], len(chroms), nb, (", %d-bp leading N block" % lead_n) if lead_n else "", cfg["n_fam"], cov,
        "ONT" if cfg["err"][1] < 0.05 else "PacBio-CLR", plan["n"], float(plan["length"].sum()) / 1e9, cfg["err"][0], cfg["err"][1], cfg["err"][2], len(g["insertions"]))
    if hard:
        import collections
        nb_k = collections.Counter(); n_k = collections.Counter()
        for f in g["hard_features"]:
            n_k[f[3]] += 1; nb_k[f[3]] += f[2] - f[1]
        text += "; HARD genome: " + ", ".join("%d %s (%.2f %% of the bases)" % (n_k[k], k, 100.0 * nb_k[k] / nb) for k in ("tandem", "micro", "lowcx", "segdup", "satellite")) + \
                ", satellites next to %d of the insertions; reads with error bursts (one per ~%d bases, %d-%d bases at %.1f x the error rates)" % (
                    n_k["satellite"], int(1 / hard["burst"][0]), hard["burst"][1], hard["burst"][2], hard["burst"][3])
    return dict(names=g["names"], ref=g["ref"], library=g["library"], reads=(buf, off, ln), read_gid=gid, loci=loci, text=text,
                total_reads=plan["n"], total_bases=int(plan["length"].sum()))


def bam_leg_run(a, rank, D, ix, qs, mo, eng, n_bases, sync, dist, device, torch, np):
    """stage 1 up to the reference's hand-off H1: reads resident -> telr_map -> coordinate-sorted, indexed BAM (TELR_alignment.py:103-114)"""
    from telr_amd.aligner import Index
    from telr_amd._abi import ALN_DTYPE
    import tempfile
    bam_dir = a.bam_dir if os.path.isdir(a.bam_dir) and os.access(a.bam_dir, os.W_OK) else tempfile.gettempdir()
    bam_path = os.path.join(bam_dir, "telr_bench_rank%d.bam" % rank)
    qnames = Index._cstr_array(["read%d" % g for g in D["read_gid"]])       # the C array of names is an input, like the reads
    tb_, to_, tl_ = concat_ref = (np.concatenate(D["ref"]), np.cumsum([0] + [len(x) for x in D["ref"]][:-1]).astype(np.int64), np.array([len(x) for x in D["ref"]], np.int32))
    if a.bam_leg == "device" and not a.no_bam_twin:
        from telr_amd._abi import MF_KEEP_CIGARS
        mo = type(mo).from_buffer_copy(mo); mo.flags |= MF_KEEP_CIGARS          # the result keeps its CIGARs on the device for the writer (what alignment() does)
    legs = []
    for rep in range(4):                      # the first pass sizes / pins the writer's buffers; of the three that follow, the median is reported
        for f in (bam_path, bam_path + ".bai"):
            if os.path.exists(f):
                os.unlink(f)
        sync()
        ix.bam_release_wait()                 # the first pass's file mapping is taken apart in the background: a run writes ONE BAM
        t0b = time.time()
        if a.bam_leg == "device" and not a.no_bam_prepare:
            ix.bam_prepare(bam_path, int((0.95 if a.bam_level else 2.9) * n_bases) + (64 << 20))
        r = ix.map_raw(qs, mo)
        t_map = time.time() - t0b
        if a.bam_leg == "host":
            ix.write_bam(r, qnames, D["reads"], D["names"], concat_ref, bam_path, md=True, cs=True, softclip=True, cmdline="bench", index=True, level=a.bam_level)
        else:
            ix.write_bam_device(r, qs, qnames, D["names"], bam_path, md=True, cs=True, softclip=True, cmdline="bench", index=True, level=a.bam_level)
        sync()
        t_all = time.time() - t0b
        n = eng.L.telr_result_count(r)
        v = np.frombuffer((ctypes.c_char * (n * ALN_DTYPE.itemsize)).from_address(eng.L.telr_result_alns(r)), dtype=ALN_DTYPE, count=n)
        ab = int(v["qlen"][(v["flags"] & 1) != 0].sum())
        ix.free_raw(r)
        legs.append((t_map, t_all, ab))
    t_map, t_all, ab = sorted(legs[1:], key=lambda x: x[1])[1]
    sz = os.path.getsize(bam_path)
    if dist is not None:
        t = torch.tensor([t_all, t_map], dtype=torch.float64, device=device if device is not None else "cpu"); dist.all_reduce(t, op=dist.ReduceOp.MAX); t_all, t_map = float(t[0]), float(t[1])
        t = torch.tensor([ab, sz], dtype=torch.float64, device=device if device is not None else "cpu"); dist.all_reduce(t); ab, sz = float(t[0]), float(t[1])
    sha = isha = None
    if a.bam_sha and os.path.exists(bam_path):
        sha = file_sha256(bam_path); isha = inflated_sha256(bam_path)
    for f in (bam_path, bam_path + ".bai"):
        if os.path.exists(f) and not os.environ.get("TELR_KEEP_BAM"):
            os.unlink(f)
    job = None
    if dist is not None and a.bam_leg == "device" and (dist.get_world_size() > 1 or a.force_exchange):
        job = job_bam_run(a, rank, D, ix, qs, mo, eng, sync, dist, device, torch, np, bam_dir)
    return {"job_bam": job, "bam_sha256": sha, "inflated_sha256": isha, "writer": a.bam_leg, "level": a.bam_level, "seconds": t_all, "map_seconds": t_map, "bam_seconds": t_all - t_map, "first_pass_seconds": legs[0][1], "seconds_of_each_pass": [x[1] for x in legs[1:]],
               "stage_ms": ix.bam_stage_ms() if a.bam_leg == "device" else None, "cigars_resident": bool(eng.L.telr_debug_bam_twin()) if a.bam_leg == "device" else None, "bam_bytes": sz, "gbp_per_s_incl_bam": ab / t_all / 1e9, "path": bam_path,
               "what": "reads resident in HBM -> telr_map -> coordinate-sorted BAM (--cs --MD -Y, SEQ + QUAL 0xff) + .bai under %s (tmpfs when that is /dev/shm: no disk in the figure); this rank's reads (at N > 1 the job's ONE file is `job_bam`); median of three passes after a first one that sizes and pins the writer's buffers" % bam_dir}


def file_sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        while True:
            b = fh.read(64 << 20)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def job_bam_run(a, rank, D, ix, qs, mo, eng, sync, dist, device, torch, np, bam_dir):
    """N > 1: ONE coordinate-sorted BAM for the job (Sniffles reads one file, TELR_sv.py:35-47), written by ALL ranks
    (shard.write_job_bam): every rank maps its reads, the records are range-partitioned by coordinate and travel -- with their
    reads as packed device words -- to the rank that owns their slice, every rank codes and writes its slice of the one file,
    rank 0 writes the one .bai."""
    from telr_amd import shard
    path = os.path.join(bam_dir, "telr_bench_job.bam")
    names = ["read%d" % g for g in D["read_gid"]]
    out = None
    world = dist.get_world_size()
    for rep in range(2):
        if rank == 0:
            for f in (path, path + ".bai"):
                if os.path.exists(f):
                    os.unlink(f)
        ix.bam_release_wait()
        dist.barrier(); sync()
        t0 = time.time()
        r = ix.map_raw(qs, mo)
        res = ix.result_arrays(r)
        t_map = time.time() - t0
        err = None
        if a.one_gpu:                  # the smoke mode puts every rank on ONE device: N grow-only mapping scratches next to the wire buffers do not fit
            eng.release_scratch()
        try:
            ph = shard.write_job_bam(path, ix, eng, res.alns, res.cigars, qs, D["reads"][2], names, D["read_gid"], D["names"], [len(x) for x in D["ref"]], dist, device,
                                     level=max(1, a.bam_level), writer_kw=dict(cmdline="bench"))
        except Exception as e:                 # (a rank that fails inside a collective takes the job down: nothing to hide here)
            err = "%s: %s" % (type(e).__name__, e); ph = {}
        aligned = int(res.alns["qlen"][(res.alns["flags"] & 1) != 0].sum())
        ix.free_raw(r)
        dist.barrier(); sync()
        t_all = time.time() - t0
    if err is not None:
        return {"error": err} if rank == 0 else None
    dev = device if device is not None else "cpu"
    t = torch.tensor([aligned], dtype=torch.float64, device=dev); dist.all_reduce(t); aligned = float(t[0])
    phases = [None] * world
    dist.all_gather_object(phases, {k: (round(v, 5) if isinstance(v, float) else v) for k, v in ph.items()})
    if rank != 0:
        return None
    out = {"records": phases[0].get("records"), "reads": int(D["total_reads"]), "aligned_bases": int(aligned), "seconds": t_all, "map_seconds": t_map,
           "write_seconds": t_all - t_map, "bam_bytes": os.path.getsize(path), "gbp_per_s_incl_bam": aligned / t_all / 1e9,
           "bam_sha256": file_sha256(path) if a.bam_sha else None, "inflated_sha256": inflated_sha256(path) if a.bam_sha else None, "path": path,
           "phase_s_per_rank": phases,
           "what": "every rank maps its reads; records range-partitioned by coordinate (sampled splitters), ONE all-to-all (%s) of records + CIGAR words + packed "
                   "2-bit reads to the owner of each coordinate slice; every rank codes the BGZF blocks of its slice on its device and writes them at its offset of the "
                   "ONE file; rank 0 writes the .bai from the gathered coordinates / virtual offsets; second of two passes" % dist.get_backend()}
    for f in (path, path + ".bai"):
        if os.path.exists(f) and not os.environ.get("TELR_KEEP_BAM"):
            os.unlink(f)
    return out


def inflated_sha256(path):
    """SHA-256 of the BGZF file's INFLATED stream (block boundaries differ between a one-rank and an N-rank file; the stream does not)"""
    import hashlib, struct, zlib
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        data = fh.read()
    o = 0
    while o < len(data):
        bsize = struct.unpack_from("<H", data, o + 16)[0] + 1
        h.update(zlib.decompress(data[o + 18:o + bsize - 8], -15))
        o += bsize
    return h.hexdigest()


def files_leg_run(a, D, eng, pname, np):
    """reads + reference as FASTA files -> telr_alignment.alignment() -> sorted BAM + .bai, timed as a whole (second of two runs)"""
    import tempfile
    from telr_amd import telr_alignment
    bam_dir = a.bam_dir if os.path.isdir(a.bam_dir) and os.access(a.bam_dir, os.W_OK) else tempfile.gettempdir()
    rf, qf, bam = (os.path.join(bam_dir, "telr_bench_" + n) for n in ("ref.fa", "reads.fa", "files.bam"))
    buf, off, ln = D["reads"]
    t0 = time.time()
    with open(rf, "wb") as fh:
        for n, r in zip(D["names"], D["ref"]):
            fh.write(b">" + n.encode() + b"\n"); fh.write(bytes(r)); fh.write(b"\n")
    # one record per read, one line per sequence: a header array, the bases and the line breaks interleaved with numpy
    hdr = [b">read%d\n" % g for g in D["read_gid"]]
    hl = np.array([len(h) for h in hdr], np.int64)
    rec = hl + ln.astype(np.int64) + 1
    pos = np.cumsum(rec) - rec
    outb = np.empty(int(rec.sum()), np.uint8)
    hb = np.frombuffer(b"".join(hdr), np.uint8)
    hflat = np.repeat(pos - (np.cumsum(hl) - hl), hl) + np.arange(len(hb)); outb[hflat] = hb
    assert (off == np.cumsum(ln.astype(np.int64)) - ln).all()          # the reads lie end to end in the buffer
    sflat = np.repeat(pos + hl - off, ln) + np.arange(int(ln.sum()))
    outb[sflat] = buf[:len(sflat)]
    outb[pos + hl + ln] = 10
    with open(qf, "wb") as fh:
        fh.write(outb.tobytes())
    del outb, hflat, sflat
    t_write = time.time() - t0
    method, presets = ("nglmr" if pname.startswith("ngmlr") else "minimap2"), ("pacbio" if pname in ("map-pb", "ngmlr-pacbio") else "ont")
    runs = []
    for rep in range(2):
        for f in (bam, bam + ".bai"):
            if os.path.exists(f):
                os.unlink(f)
        eng.L.telr_bam_release_wait()
        t0 = time.time()
        telr_alignment.alignment(bam, qf, rf, bam_dir, "bench", 1, method, presets, engine=eng)
        runs.append(time.time() - t0)
        phases = dict(getattr(telr_alignment.alignment, "last_timings", {}))
    sz = os.path.getsize(bam)
    nb = int(ln.sum())
    for f in (rf, qf, bam, bam + ".bai"):
        if os.path.exists(f) and not os.environ.get("TELR_KEEP_BAM"):
            os.unlink(f)
    return {"seconds": runs[-1], "first_run_seconds": runs[0], "read_bases": nb, "gbp_per_s_from_files": nb / runs[-1] / 1e9, "phases_s": phases, "bam_bytes": sz,
            "fasta_written_in_s": t_write, "call": "telr_alignment.alignment(bam, reads.fa, ref.fa, out, sample, thread, %r, %r)" % (method, presets),
            "what": "FASTA files under %s (page cache warm) -> telr_fasta_load x2, pack + upload, index build, telr_map, telr_write_bam_dev -> sorted BAM + .bai: the wall clock of the reference's stage 1 (TELR_alignment.py:9-114)" % bam_dir}


def save_dataset(path, D):
    """arrays + a JSON blob, no pickle (np.load(..., allow_pickle=False) on the way back)"""
    import numpy as np
    arr = {"ref_%d" % i: r for i, r in enumerate(D["ref"])}
    arr.update({"lib_%d" % i: np.asarray(r) for i, r in enumerate(D["library"])})
    arr["reads_buf"], arr["reads_off"], arr["reads_len"] = D["reads"]
    arr["read_gid"] = np.asarray(D["read_gid"])
    loci_arr = {}
    loci = []
    for i, l in enumerate(D["loci"]):
        m = {}
        for k, v in l.items():
            if isinstance(v, np.ndarray):
                loci_arr["locus_%d_%s" % (i, k)] = v; m[k] = {"__arr__": "locus_%d_%s" % (i, k)}
            elif isinstance(v, (bytes, bytearray)):
                loci_arr["locus_%d_%s" % (i, k)] = np.frombuffer(bytes(v), np.uint8); m[k] = {"__bytes__": "locus_%d_%s" % (i, k)}
            else:
                m[k] = v
        loci.append(m)
    arr.update(loci_arr)
    meta = {"names": list(D["names"]), "n_ref": len(D["ref"]), "n_lib": len(D["library"]), "loci": loci, "text": D["text"],
            "total_reads": int(D["total_reads"]), "total_bases": int(D["total_bases"])}
    arr["meta_json"] = np.frombuffer(json.dumps(meta, default=lambda o: o.item() if hasattr(o, "item") else str(o)).encode(), np.uint8)
    tmp = path + ".tmp.npz"
    np.savez(tmp, **arr)
    os.replace(tmp, path)


def load_dataset(path):
    import numpy as np
    z = np.load(path, allow_pickle=False)
    meta = json.loads(bytes(z["meta_json"]).decode())
    loci = []
    for m in meta["loci"]:
        l = {}
        for k, v in m.items():
            if isinstance(v, dict) and "__arr__" in v:
                l[k] = z[v["__arr__"]]
            elif isinstance(v, dict) and "__bytes__" in v:
                l[k] = bytes(z[v["__bytes__"]])
            else:
                l[k] = v
        loci.append(l)
    return dict(names=meta["names"], ref=[z["ref_%d" % i] for i in range(meta["n_ref"])], library=[z["lib_%d" % i] for i in range(meta["n_lib"])],
                reads=(z["reads_buf"], z["reads_off"], z["reads_len"]), read_gid=z["read_gid"], loci=loci, text=meta["text"],
                total_reads=meta["total_reads"], total_bases=meta["total_bases"])


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a))
    cfg = CONFIGS[a.config]
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    # stdout carries the ONE JSON line and nothing else: libraries write banners to file descriptor 1 (RCCL prints its
    # version block there when the communicator is created, gloo its own), so the descriptor is pointed at stderr for the
    # duration of the run and the line goes out through a private copy of the original
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    local = 0 if a.one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    # ranks of one node share its CPUs: each engine sizes its host pool for its share (the library's default is 1.5 x the
    # CPUs the process may use, which every rank would claim for itself)
    lws = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if lws > 1 and "TELR_HOST_THREADS" not in os.environ:
        os.environ["TELR_HOST_THREADS"] = str(max(4, min(48, usable_cpus() * 3 // (2 * lws))))
    if a.dry_launch:
        import torch.distributed as dist
        dist.init_process_group(a.backend)
        import torch
        t = torch.tensor([rank + 1], dtype=torch.int64)
        if a.backend == "nccl":
            torch.cuda.set_device(local); t = t.cuda()
        dist.all_reduce(t)
        if rank == 0:
            json_out.write(json.dumps({"dry_launch": True, "n_gpus": world, "rank_sum": int(t.item()), "backend": a.backend}) + "\n"); json_out.flush()
        dist.destroy_process_group()
        return
    import numpy as np
    t0 = time.time()
    D = None
    cache = ""
    if a.data_cache:
        import hashlib
        src = open(os.path.join(ROOT, "telr_amd", "synth.py"), "rb").read()
        keytxt = repr((a.config, a.scaling, rank, world, a.coverage, a.genome_scale, a.genome_len, a.reads, a.read_bases, a.insertions, sorted(cfg.items()))).encode() + src
        cache = os.path.join(a.data_cache, "telr_bench_%s_r%dof%d_%s.npz" % (a.config, rank, world, hashlib.sha256(keytxt).hexdigest()[:16]))
        os.makedirs(a.data_cache, mode=0o700, exist_ok=True)
        if os.path.exists(cache):
            D = load_dataset(cache)
        elif a.require_cache:
            sys.exit("bench.py: --require-cache and %s does not exist" % cache)
    if D is None:
        D = build_dataset(a, cfg, rank, world, lws)       # CPU only; forks workers: before any GPU initialisation
        if cache:
            save_dataset(cache, D)
    t_gen = time.time() - t0

    import torch
    dist = None
    device = None
    if world > 1 or "RANK" in os.environ:      # launched by torch.distributed.run (also at world size 1)
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            torch.cuda.set_device(local)
            device = torch.device("cuda", local)
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(a.backend)
    from telr_amd.aligner import Engine, Index, SeqSet, _np_from
    from telr_amd._abi import ALN_DTYPE
    from telr_amd.presets import preset
    from telr_amd import telr_assembly, locus_pipeline, shard

    pname = a.preset or cfg["preset"]
    io, mo = preset(pname)
    if a.fill_band_q4:
        mo.fill_band_q4 = a.fill_band_q4
    for kv in filter(None, a.map_opt.split(",")):
        k_, v_ = kv.split("=")
        setattr(mo, k_, type(getattr(mo, k_))(float(v_)))
    ref_strs = [bytes(r).decode() for r in D["ref"]]
    eng = Engine(local)
    t0 = time.time()
    ix = eng.index(ref_strs, io)
    t_index = time.time() - t0
    t0 = time.time()
    qs = eng.seqset(D["reads"])           # 2-bit packing on the host + H2D of this rank's whole read set
    t_upload_first = time.time() - t0     # includes pinning the context's staging buffers (once per context)
    t0 = time.time()
    qs2 = eng.seqset(D["reads"])          # the steady state of a run that streams read batches: staging already pinned
    t_upload = time.time() - t0
    qs2.free()
    n_bases = qs.bases()

    # Steps are streamed the way a stage-1 run over many read batches would be: telr_map returns when the alignment
    # records are complete, the DMA of the last range's CIGARs finishes in the background, and a result is released only
    # after the next call has been issued.  Everything, the last DMA included, is complete before the closing synchronisation.
    def sync():
        torch.cuda.synchronize(local)
        if dist is not None:
            dist.barrier()

    def timed_map(ix_, mo_):
        """2 set-up calls + a.warmup untimed + EXACTLY a.steps timed steps of ix_.map_raw(qs, mo_), bracketed by sync(); max over ranks"""
        held = []

        def step():
            r = ix_.map_raw(qs, mo_)
            L = eng.L
            n = L.telr_result_count(r)
            # a VIEW of the library-owned record array (valid until the result is freed: after the next step): counting the
            # aligned bases must not copy 50 MB of records per step inside the timed region
            al = np.frombuffer((ctypes.c_char * (n * ALN_DTYPE.itemsize)).from_address(L.telr_result_alns(r)), dtype=ALN_DTYPE, count=n) if n else np.zeros(0, ALN_DTYPE)
            while held:
                ix_.free_raw(held.pop())
            held.append(r)
            return int(al["qlen"][(al["flags"] & 1) != 0].sum()), al

        step(); step()      # set-up, not warm-up steps: the first calls size the engine's grow-only scratch and pin the TWO
                            # result buffers that are alive at any time when steps are streamed
        for _ in range(a.warmup):
            step()
        T = dict(stage_tot={}, cls_tot=None, ctr_tot={}, launches=0, retries=0)
        sync()
        t0 = time.time()
        aligned = 0
        for _ in range(a.steps):
            b, al = step()
            aligned += b
            for k, v in eng.stage_ms().items():
                T["stage_tot"][k] = T["stage_tot"].get(k, 0.0) + v
            c = eng.dp_classes(); T["cls_tot"] = c if T["cls_tot"] is None else T["cls_tot"] + c
            for k, v in eng.counters().items():
                T["ctr_tot"][k] = T["ctr_tot"].get(k, 0) + v
            T["launches"] += int(eng.L.telr_debug_pk_launches(eng.h)); T["retries"] += int(eng.L.telr_debug_dp_retries(eng.h))
        if held:
            eng.L.telr_result_wait(held[-1])      # the last step's CIGAR array is home as well
        sync()
        dt = time.time() - t0
        T["dt_local"] = dt
        T["al"] = al.copy()                       # the last step's records outlive the result handle (window reads of the loci leg)
        while held:
            ix_.free_raw(held.pop())
        if dist is not None:
            dev = device if device is not None else "cpu"
            t = torch.tensor([dt], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t[0])
            s_ = torch.tensor([aligned], dtype=torch.float64, device=dev); dist.all_reduce(s_, op=dist.ReduceOp.SUM); aligned = float(s_[0])
        T["dt"] = dt; T["aligned"] = aligned; T["value"] = aligned / dt / 1e9
        return T

    PK = list(range(10, 18)) + [22]

    def pk_roofline(T):
        """the contract's roofline object of k_dp_pk for one timed leg: live HIP-event launch time (telr_stage_ms "k_dp_pk") and the
        algorithmic bytes the library counted per problem (2-bit query + target bases once, 4 B per CIGAR run, 32 B result)"""
        cls = T["cls_tot"]
        launches = max(1, T["launches"])
        k_ms_tot = T["stage_tot"].get("k_dp_pk", 0.0)
        k_bytes_tot = float(cls[PK, 3].sum())
        achieved = k_bytes_tot / (k_ms_tot * 1e-3) / 1e9 if k_ms_tot > 0 else 0.0
        return {"bound": "hbm", "kernel": "k_dp_pk", "dp_classes": PK, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "launch_ms": k_ms_tot / launches, "launches_per_step": launches / a.steps, "algorithmic_bytes_per_launch": k_bytes_tot / launches,
                "problems_per_launch": float(cls[PK, 0].sum()) / launches, "cells_per_launch": float(cls[PK, 1].sum()) / launches,
                "gcups": float(cls[PK, 1].sum()) / (k_ms_tot * 1e-3) / 1e9 if k_ms_tot > 0 else None}

    T0 = timed_map(ix, mo)
    stage_tot, cls_tot, ctr_tot, launches, retries = T0["stage_tot"], T0["cls_tot"], T0["ctr_tot"], T0["launches"], T0["retries"]
    dt, dt_local, aligned, al, value = T0["dt"], T0["dt_local"], T0["aligned"], T0["al"], T0["value"]
    per_rank_ms = [dt_local / a.steps * 1e3]
    if dist is not None:
        dev = device if device is not None else "cpu"
        t = torch.tensor([t_upload], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); t_upload_max = float(t[0])
        s = torch.tensor([n_bases], dtype=torch.float64, device=dev); dist.all_reduce(s, op=dist.ReduceOp.SUM); job_bases = float(s[0])
        pr = torch.zeros(world, dtype=torch.float64, device=dev); pr[rank] = dt_local / a.steps * 1e3; dist.all_reduce(pr); per_rank_ms = [float(x) for x in pr.cpu()]
    else:
        t_upload_max, job_bases = t_upload, float(n_bases)
    value_h2d = aligned / (dt + a.steps * t_upload_max) / 1e9

    # ---- the reference's DEFAULT stage-1 aligner on the same reads (TELR_input.py:176-177 `--aligner nglmr`, TELR_alignment.py:28-51):
    # the same measurement -- same read set resident, same warm-up, same a.steps timed steps, same synchronisation -- with the
    # `ngmlr-ont` / `ngmlr-pacbio` preset on its own (w,k) = (5,13) index.  Skipped when the headline preset already is that one.
    default_aligner = None
    pname_ng = "ngmlr-ont" if cfg["err"][1] < 0.05 else "ngmlr-pacbio"
    ix_ng = None
    if not a.no_default_aligner_leg and pname != pname_ng and not a.map_opt:
        try:
            eng.release_scratch()
            io_ng, mo_ng = preset(pname_ng)
            t0 = time.time()
            ix_ng = eng.index(ref_strs, io_ng)
            t_index_ng = time.time() - t0
            T1 = timed_map(ix_ng, mo_ng)
            al1 = T1["al"]
            rf = pk_roofline(T1)
            dp_ms1 = T1["stage_tot"].get("dp", 0.0) / a.steps
            rf["note"] = "same definition as `roofline`; all DP kernels together: %.1f ms per step" % dp_ms1
            rf["traffic"] = None
            try:            # HBM bytes per launch from the committed PMC passes of the same preset on this workload (rocprofv3 --pmc cannot run inside this process)
                pmn = json.load(open(os.path.join(ROOT, "profiles", PMC_PROFILE_NG)))
                if pmn.get("config") == a.config and pname_ng == "ngmlr-ont" and not a.coverage and a.genome_scale == 1.0 and world == 1:
                    rf["traffic"] = pmn["traffic_bytes_per_launch"]
                    rf["from_profile"] = {"file": "profiles/" + PMC_PROFILE_NG, "collected_at_commit": pmn.get("head_sha"), "traffic_over_algorithmic": pmn.get("traffic_over_algorithmic"), "valu_issue": pmn.get("valu_issue")}
            except Exception:
                pass
            default_aligner = {"preset": pname_ng, "value": T1["value"], "unit": "Gbp/s", "ms_per_step": T1["dt"] / a.steps * 1e3, "steps": a.steps, "warmup": a.warmup,
                               "index_build_s": t_index_ng, "roofline": rf,
                               "stage_ms_per_step": {k: v / a.steps for k, v in T1["stage_tot"].items()},
                               "dp_classes": {str(c): [int(x) // a.steps for x in T1["cls_tot"][c]] for c in range(T1["cls_tot"].shape[0]) if T1["cls_tot"][c, 0]},
                               "counters": {k: v / a.steps for k, v in T1["ctr_tot"].items()}, "dp_retries": T1["retries"] // a.steps,
                               "frac_reads_mapped": len(np.unique(al1["qid"][(al1["flags"] & 1) != 0])) / max(1, len(D["reads"][2])),
                               "what": "the reference's default stage-1 aligner (`--aligner nglmr` -> `ngmlr -x %s`, TELR_input.py:176-177, TELR_alignment.py:28-51) on the SAME resident read set: "
                                       "preset %s (13-mers at every third position, sub-read voting, NGMLR's convex gap cost exactly), timed exactly like `value`" % (pname_ng.split("-")[1], pname_ng)}
            del T1, al1
            eng.release_scratch()
        except Exception as e:
            default_aligner = {"preset": pname_ng, "error": "%s: %s" % (type(e).__name__, e)}

    # ---- host-inclusive rate of a run that STREAMS its read batches: while step k maps, a second context packs and uploads
    # the reads of step k+1 (the same read set again, from the caller's ASCII buffer).  Timed from before the first upload to
    # the last CIGAR word; `value` above stays the resident-input rate the metric is quoted on.
    value_stream = None
    if not a.no_stream_leg:
        import threading
        eng2 = Engine(local)
        eng2.seqset(D["reads"]).free()            # pins that context's staging (once per context, as for the first)
        sync()
        t0s = time.time()
        cur = eng2.seqset(D["reads"])
        nxt = [None]

        def upload():
            nxt[0] = eng2.seqset(D["reads"])
        aligned_s = 0
        prev = None
        for k in range(a.steps):
            th = threading.Thread(target=upload) if k + 1 < a.steps else None
            if th:
                th.start()
            r = ix.map_raw(cur, mo)
            n = eng.L.telr_result_count(r)
            v = np.frombuffer((ctypes.c_char * (n * ALN_DTYPE.itemsize)).from_address(eng.L.telr_result_alns(r)), dtype=ALN_DTYPE, count=n) if n else np.zeros(0, ALN_DTYPE)
            aligned_s += int(v["qlen"][(v["flags"] & 1) != 0].sum())
            if prev is not None:
                ix.free_raw(prev)
            prev = r
            if th:
                th.join()
                cur.free(); cur = nxt[0]
        eng.L.telr_result_wait(prev)
        sync()
        dts = time.time() - t0s
        ix.free_raw(prev); cur.free(); eng2.close()
        if dist is not None:
            t = torch.tensor([dts], dtype=torch.float64, device=device if device is not None else "cpu"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dts = float(t[0])
            t = torch.tensor([aligned_s], dtype=torch.float64, device=device if device is not None else "cpu"); dist.all_reduce(t); aligned_s = float(t[0])
        value_stream = aligned_s / dts / 1e9

    # ---- what one rank of an N-rank strong-scaling run would do: the shard of rank 0 at N = 2 / 4 / 8, on this GPU -----------------
    shard_out = None
    if world == 1 and not a.no_shard_leg and a.scaling == "strong" and len(D["reads"][2]) >= 4096:
        try:
            shard_out = {"what": "the reads rank 0 of an N-rank run is dealt (shard.shard_reads: by cumulative bases), mapped alone on this GPU, 6 streamed steps after two set-up calls (as `value` is timed; rounds 1-4 released every result before the next call, i.e. waited for its CIGAR DMA: 0.73 instead of 0.78 at 8 ranks): "
                                 "an upper bound of the strong-scaling efficiency (nothing else of an N-rank node interferes here)", "ranks": {}}
            ln_all = D["reads"][2]
            for nrk in (2, 4, 8):
                idx = np.array(shard.shard_reads(ln_all, nrk)[0], np.int32)
                sub = qs.subset(idx)
                rr = ix.map_raw(sub, mo); ix.free_raw(rr)
                rr = ix.map_raw(sub, mo); ix.free_raw(rr)
                sync(); t0s = time.time(); ab = 0; prev_ = None
                n_sh = 6
                for _ in range(n_sh):          # streamed exactly as the timed steps of `value` are: a result is released after the next call has been issued
                    rr = ix.map_raw(sub, mo)
                    n_ = eng.L.telr_result_count(rr)
                    v_ = np.frombuffer((ctypes.c_char * (n_ * ALN_DTYPE.itemsize)).from_address(eng.L.telr_result_alns(rr)), dtype=ALN_DTYPE, count=n_)
                    ab += int(v_["qlen"][(v_["flags"] & 1) != 0].sum())
                    if prev_ is not None:
                        ix.free_raw(prev_)
                    prev_ = rr
                eng.L.telr_result_wait(prev_)
                sync(); dts_ = time.time() - t0s
                ix.free_raw(prev_)
                rate = ab / dts_ / 1e9
                shard_out["ranks"][str(nrk)] = {"shard_gbp": float(ln_all[idx].sum()) / 1e9, "ms_per_step": dts_ / n_sh * 1e3, "gbp_per_s": rate, "efficiency_vs_one_gpu": rate / value,
                                                "stage_ms_last_call": {k: round(v, 2) for k, v in eng.stage_ms().items() if v >= 0.3}}
                # round 6: what the shard's BAM costs its rank locally -- records, sort, BGZF on the device and the write of the rank's
                # share of the file (tmpfs) -- so that the estimate below is end to end, not map-only.  (Not in it: the record exchange, 0.8 GB per
                # rank over xGMI at 8 ranks, and rank skew: no second device on this pool.)
                if a.bam_leg == "device":
                    try:
                        from telr_amd._abi import MF_KEEP_CIGARS
                        from telr_amd.aligner import Index
                        import tempfile
                        mo_k = type(mo).from_buffer_copy(mo); mo_k.flags |= MF_KEEP_CIGARS
                        bdir = a.bam_dir if os.path.isdir(a.bam_dir) and os.access(a.bam_dir, os.W_OK) else tempfile.gettempdir()
                        bpath = os.path.join(bdir, "telr_bench_shard%d.bam" % nrk)
                        qn_sub = Index._cstr_array(["read%d" % D["read_gid"][i] for i in idx])
                        tb_s = []
                        sh_bases = int(ln_all[idx].sum())
                        for rep in range(4):               # (as the stage-1-to-BAM leg: the file prepared in the background while the reads map; the first pass sizes the writer)
                            for f in (bpath, bpath + ".bai"):
                                if os.path.exists(f):
                                    os.unlink(f)
                            ix.bam_release_wait(); sync(); t0b = time.time()
                            if not a.no_bam_prepare:
                                ix.bam_prepare(bpath, int((0.95 if a.bam_level else 2.9) * sh_bases) + (64 << 20))
                            rk = ix.map_raw(sub, mo_k); t_m = time.time() - t0b
                            ix.write_bam_device(rk, sub, qn_sub, D["names"], bpath, md=True, cs=True, softclip=True, cmdline="bench", index=True, level=a.bam_level)
                            sync(); tb_s.append(time.time() - t0b - t_m)
                            ix.free_raw(rk)
                        for f in (bpath, bpath + ".bai"):
                            if os.path.exists(f):
                                os.unlink(f)
                        shard_out["ranks"][str(nrk)]["bam_seconds_local"] = sorted(tb_s[1:])[1]          # median of the three passes behind the first
                    except Exception as e:
                        shard_out["ranks"][str(nrk)]["bam_seconds_local"] = None; shard_out["ranks"][str(nrk)]["bam_error"] = "%s: %s" % (type(e).__name__, e)
                sub.free()
        except Exception as e:
            shard_out = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- stage 1 up to the reference's hand-off H1: a coordinate-sorted, indexed BAM (TELR_alignment.py:103-114) ------------
    bam_out = None
    if a.bam_leg != "none":
        try:
            bam_out = bam_leg_run(a, rank, D, ix, qs, mo, eng, n_bases, sync, dist, device, torch, np)
        except Exception as e:           # the leg is an extra: a full /dev/shm or a device without room for it must not cost the bench line
            bam_out = {"writer": a.bam_leg, "error": "%s: %s" % (type(e).__name__, e)}

    # ---- stage 1 from files to the sorted BAM: what `telr` itself would wait for (TELR_alignment.py:9-114) ---------------------
    files_out = None
    if a.files_leg and world == 1:
        try:
            files_out = files_leg_run(a, D, eng, pname, np)
        except Exception as e:
            files_out = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- TE loci/s: the per-locus bundle on window reads taken from THIS run's stage-1 records -----------------------
    loci_out = None
    n_loci = len(D["loci"]) if a.loci < 0 else min(a.loci, len(D["loci"]))
    if n_loci > 0:
        loci = D["loci"][:n_loci]
        # stage 1 is over: its grow-only mapping scratch (150-200 GB for a 30x set, two slots) and the BAM writer's buffers go back
        # before the per-locus stages size their own (in the pipeline, Sniffles and the assemblers run between the two)
        eng.release_scratch()
        io10, _ = preset("asm10")
        ix10 = eng.index(ref_strs, io10)
        lib_names = ["fam%d" % i for i in range(len(D["library"]))]
        lib = [bytes(x).decode() for x in D["library"]]
        chrom_ids = {n: i for i, n in enumerate(D["names"])}
        ref_of = dict(zip(D["names"], ref_strs))
        presets_arg = "ont" if cfg["err"][1] < 0.05 else "pacbio"
        rbuf, roff, rln = D["reads"]
        # LPT cost of a locus = what its bundle maps: the contig and ALT bases plus the bases of its window reads TWICE (S6 maps every window
        # read against the forward and the reverse-complement contig) -- the window reads are most of it and differ five-fold between loci.
        # At N > 1 a rank sees the window reads of its own share of the reads: one all-reduce of n_loci counters (metadata, not on the data path).
        wr0 = telr_assembly.window_reads(al, chrom_ids, [(l["chrom"], l["start"], l["end"]) for l in loci])
        rb_loc = np.array([int(rln[w].sum()) for w in wr0], np.int64)
        if dist is not None and world > 1:
            t_ = torch.from_numpy(rb_loc).to(device if device is not None else "cpu"); dist.all_reduce(t_); rb_loc = t_.cpu().numpy()
        shards = shard.shard_loci([locus_pipeline.locus_cost(l) + 2 * int(b) for l, b in zip(loci, rb_loc)], world)
        del wr0
        owner = {}
        for r_, lst in enumerate(shards):
            for i in lst:
                owner[i] = r_

        phase = {}                                 # seconds per phase of the LAST pass (exchange, bundle, all-gather)

        def loci_pass(polish=None):
            phase.clear()
            # a12: every read with ANY stage-1 record overlapping [bp-1000, bp+1000) (TELR_assembly.py:384-415), from the
            # records of the last step (this rank's reads)
            wr = telr_assembly.window_reads(al, chrom_ids, [(l["chrom"], l["start"], l["end"]) for l in loci])
            if world == 1 and not (a.force_exchange and dist is not None):
                for l, idx in zip(loci, wr):
                    l["read_idx"] = idx.astype(np.int32)
                return locus_pipeline.run_loci_distributed(eng, ix10, D["names"], lambda ch: ref_of[ch], loci, lib_names, lib, shards=shards,
                                                           presets=presets_arg, read_set=qs, timings=phase, polish=polish)
            t_ex = time.time()
            lid = np.repeat(np.arange(len(wr)), [len(x) for x in wr]); ridx = np.concatenate(wr) if len(wr) else np.zeros(0, np.int64)
            own = np.array([owner[i] for i in range(len(loci))], np.int64)
            # the window reads travel as what they are on the device: 2-bit words + ambiguity mask of the resident read set,
            # gathered by one kernel in destination order, ONE all-to-all of int32 words (device tensors under RCCL), and the
            # pooled set is built from the received words in place (telr_seqset_from_packed): no ASCII, no host staging
            held = []

            def gather_packed(order):
                sub = qs.subset(ridx[order]); held.append(sub)
                return sub.packed()
            g_loc, g_read, g_len, w2, wn, order = shard.exchange_window_reads_packed(lid, D["read_gid"][ridx], own[lid], rln[ridx], gather_packed, dist, device, timings=phase)
            pool_set = SeqSet.from_packed(eng, g_len, w2, wn)
            for sub in held:
                sub.free()
            cuts = np.searchsorted(g_loc[order], np.arange(len(loci) + 1))
            for li, l in enumerate(loci):
                l["read_idx"] = order[cuts[li]:cuts[li + 1]].astype(np.int32)      # places in the pool, by read id
                l.pop("reads", None)
            phase["exchange_s"] = phase.get("exchange_s", 0.0) + time.time() - t_ex
            return locus_pipeline.run_loci_distributed(eng, ix10, D["names"], lambda ch: ref_of[ch], loci, lib_names, lib, dist=dist, device=device,
                                                       shards=shards, presets=presets_arg, read_set=pool_set, timings=phase, polish=polish)
        flank_parity = None
        if a.flank_parity and rank == 0:
            from telr_amd.fasta import concat
            io10b, mo10 = preset("asm10"); mo10.best_n = 10
            fl = []
            for l in loci:
                c = l["contig"] if isinstance(l["contig"], (bytes, bytearray)) else l["contig"].encode()
                fl.append(np.frombuffer(c[:500], np.uint8)); fl.append(np.frombuffer(c[-500:], np.uint8))
            sys.stderr.write("[site parity] S7 flanks...\n"); sys.stderr.flush()
            flank_parity = cpu_baseline(ref_strs, concat(fl), io10b, mo10, len(fl), gpu_index=ix10, seed=0)
            sys.stderr.write("[site parity] S7 done; S6 oracle...\n"); sys.stderr.flush()
            # S6 the same way: the window reads of the first 150 loci against the forward and the reverse-complement contig of their
            # locus (per-query targets), map-ont / map-pb, engine vs oracle
            from oracle import binding as ob
            from telr_amd.fasta import revcomp
            sub_loci = loci[:150]
            wr6 = telr_assembly.window_reads(al, chrom_ids, [(l["chrom"], l["start"], l["end"]) for l in sub_loci])
            tg6 = []
            for l in sub_loci:
                c = l["contig"] if isinstance(l["contig"], str) else bytes(l["contig"]).decode()
                tg6 += [c, revcomp(c)]
            q6, qt_fw = [], []
            for k, w in enumerate(wr6):
                for i in w:
                    q6.append(bytes(rbuf[roff[i]:roff[i] + rln[i]]).decode()); qt_fw.append(2 * k)
            qt6 = np.array(qt_fw + [x + 1 for x in qt_fw], np.int32); q6 = q6 + q6
            io6, mo6 = preset("map-ont" if presets_arg == "ont" else "map-pb")
            t06 = time.time()
            o6 = ob.OracleIndex(tg6, io6).map(q6, mo6, qtarget=qt6)
            t_or = time.time() - t06
            sys.stderr.write("[site parity] S6 oracle done; S6 engine...\n"); sys.stderr.flush()
            e6 = eng.index(tg6, io6).map(q6, mo6, qtarget=qt6)
            sys.stderr.write("[site parity] S6 engine done\n"); sys.stderr.flush()
            w6, g6 = _read_digests(o6["alns"], o6["cigars"]), _read_digests(e6.alns, e6.cigars)
            bad6 = [q for q in set(w6) | set(g6) if w6.get(q) != g6.get(q)]
            # and the whole per-locus bundle -- S4, S5, S6 + depth + allele frequency, S7 + the liftover tree -- on the first 60 loci,
            # driven once by the engine and once by the oracle behind the same host code (tests/oracle_backend.py)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from oracle_backend import OracleBackend
            b_loci = []
            for l, w in zip(loci[:60], wr6[:60]):
                b_loci.append(dict(l, reads=[bytes(rbuf[roff[i]:roff[i] + rln[i]]).decode() for i in w]))
                b_loci[-1].pop("read_idx", None)
            if os.environ.get("TELR_DUMP_BUNDLE"):
                import pickle
                with open(os.environ["TELR_DUMP_BUNDLE"], "wb") as fh:
                    pickle.dump({"loci": [{k: (v if isinstance(v, (str, int, float, list, dict, tuple)) else bytes(v)) for k, v in l.items()} for l in b_loci], "lib_names": lib_names, "lib": lib, "presets": presets_arg}, fh)
            t0b_ = time.time()
            obk = OracleBackend()
            bo = locus_pipeline.run_loci(obk, obk.index(ref_strs, io10b), D["names"], lambda ch: ref_of[ch], b_loci, lib_names, lib, presets=presets_arg)
            t_bo = time.time() - t0b_
            sys.stderr.write("[site parity] bundle oracle done; bundle engine...\n"); sys.stderr.flush()
            bh = locus_pipeline.run_loci(eng, ix10, D["names"], lambda ch: ref_of[ch], b_loci, lib_names, lib, presets=presets_arg)
            flank_parity["bundle"] = {"loci": len(b_loci), "annotation_rows": len(bh["annotation"]), "liftover_reports": len(bh["liftover"]), "af_entries": len(bh["af"]),
                                      "identical": bh["annotation"] == bo["annotation"] and bh["liftover"] == bo["liftover"] and bh["summary"] == bo["summary"] and bh["af"] == bo["af"],
                                      "oracle_seconds": t_bo,
                                      "what": "locus_pipeline.run_loci on the first 60 loci (their real window reads, the full-size reference for S7): annotation rows, liftover reports, summary and allele-frequency tables of the engine run == those of the oracle run"}
            flank_parity["s6"] = {"loci": len(sub_loci), "queries": len(q6), "query_bases": int(sum(len(x) for x in q6)), "records_oracle": int(len(o6["alns"])), "records_engine": int(len(e6.alns)),
                                  "queries_differing": len(bad6), "identical": not bad6, "oracle_seconds_one_thread": t_or,
                                  "what": "call site S6: every window read of the first 150 loci against the forward and the reverse-complement contig of its locus (qtarget), all records and CIGARs, engine vs CPU oracle"}
        loci_pass()                                # warm-up (sizes the scratch)
        sync()
        prof = None
        if os.environ.get("TELR_PROF_LOCI") and rank == 0:      # where does the per-locus leg spend its time? (stderr)
            import cProfile
            prof = cProfile.Profile(); prof.enable()
        # The process holds millions of long-lived Python objects by now (458,550 read names, the loci, the data set): a full
        # collection of the cyclic garbage collector walks all of them -- 85 ms, every fourth pass.  They are moved to the
        # permanent generation, as a long-running pipeline would do after loading its inputs.
        import gc
        gc.collect(); gc.freeze()
        # three timed passes, the median reported: one pass is 120-160 ms of four small engine calls and Python, and a single
        # sample of it swings by 25 % from run to run (every pass gives the same table)
        t_passes = []
        free_first = eng.mem_info()[0]
        for _ in range(1 if prof is not None else int(os.environ.get("TELR_LOCI_PASSES", "3"))):
            if dist is not None:
                dist.barrier()
            sync()
            t0 = time.time()
            rows, lres = loci_pass()
            sync()
            t_passes.append(time.time() - t0)
        t_loci = sorted(t_passes)[len(t_passes) // 2]
        free_last = eng.mem_info()[0]
        phase_all = [{k: round(v, 5) for k, v in phase.items()}]
        if dist is not None and world > 1:
            got = [None] * world
            dist.all_gather_object(got, phase_all[0])
            phase_all = got
        if prof is not None:
            import pstats
            prof.disable()
            st = pstats.Stats(prof, stream=sys.stderr); st.sort_stats("cumulative").print_stats(40); st.sort_stats("tottime").print_stats(25)
        if dist is not None:
            t = torch.tensor([t_loci], dtype=torch.float64, device=device if device is not None else "cpu"); dist.all_reduce(t, op=dist.ReduceOp.MAX); t_loci = float(t[0])
        fam_of = {i: n for i, n in enumerate(lib_names)}

        def tally(rows):
            good = 0; af_ok = 0
            why = {"no row (no TE annotated on the contig)": n_loci - len(rows), "annotation merged with a neighbouring reference TE copy (several families)": 0,
                   "unlifted (flanks not placed next to each other)": 0, "other": 0}
            for r in rows:
                l = loci[int(r["locus_id"])]; tr = l["truth"]
                if r["type"] == 1 and r["chrom_id"] == chrom_ids[tr["chrom"]] and abs(int(r["start"]) - tr["pos"]) <= 20 and \
                        (1 if tr["strand"] == "+" else -1) == r["strand"] and r["n_family"] >= 1 and fam_of[int(r["family_id"][0])] == tr["family"]:
                    good += 1
                    if not np.isnan(r["af"]) and abs(float(r["af"]) - tr["af"]) <= 0.15:
                        af_ok += 1
                elif r["n_family"] > 1:
                    why["annotation merged with a neighbouring reference TE copy (several families)"] += 1
                elif r["type"] == 0:
                    why["unlifted (flanks not placed next to each other)"] += 1
                else:
                    why["other"] += 1
            return good, af_ok, why
        n_rows = len(rows)
        good, af_ok, why = tally(rows)
        # round 6: the loci leg at 1 / N of the load -- the loci rank 0 of an N-rank run is dealt (LPT), run alone on this GPU
        loci_shard_s = {}
        if world == 1 and shard_out is not None and "ranks" in shard_out:
            try:
                costs = [locus_pipeline.locus_cost(l) + 2 * int(b) for l, b in zip(loci, rb_loc)]
                for nrk in (2, 4, 8):
                    mine = shard.shard_loci(costs, nrk)[0]
                    lsub = [loci[i] for i in mine]
                    ts_ = []
                    for rep in range(3):
                        sync(); t0 = time.time()
                        locus_pipeline.run_loci_distributed(eng, ix10, D["names"], lambda ch: ref_of[ch], lsub, lib_names, lib, shards=[list(range(len(lsub)))], presets=presets_arg, read_set=qs)
                        sync(); ts_.append(time.time() - t0)
                    loci_shard_s[str(nrk)] = {"loci": len(lsub), "seconds": sorted(ts_)[1]}
            except Exception as e:
                loci_shard_s = {"error": "%s: %s" % (type(e).__name__, e)}
        # The OTHER call set, in every line: the per-locus calls S4-S6 run minimap2's `-x map-ont|map-pb`, whose 2.22 defaults
        # carry the long join (-r500,20000) -- [recall]-grade, there is no minimap2 here to settle it.  With the long join a library hit
        # chains ACROSS an insertion nested in a reference TE copy and the reference's own merge / decision tree calls the locus
        # "reference" (DESIGN 3.11); without it the hit breaks at the insertion.  Same loci, same reads, bw_long = 0 in every per-locus preset:
        alt_set = None
        try:
            from telr_amd import presets as _presets
            with _presets.override(bw_long=0):
                rows0, _ = loci_pass()
            g0, a0, w0 = tally(rows0)
            alt_set = {"recovered_exact_chrom_family_strand_pos20": g0, "of_those_af_within_0.15": a0, "rows_in_merged_table": len(rows0), "not_recovered": w0,
                       "what": "the same pass with bw_long = 0 in the map-ont / map-pb presets of S4-S6 (the round-2 records: a library hit ends at a nested insertion)"}
        except Exception as e:
            alt_set = {"error": "%s: %s" % (type(e).__name__, e)}
        # the polishing hand-off (H3) with the consensus made on the device (telr_consensus_build: pile-up majority vote, NOT
        # wtpoa-cns's POA -- an extra behind a flag of the pipeline, timed here once, outside the loci/s figure)
        polish = None
        if world == 1 and not a.no_polish_leg:
            try:
                wr_p = telr_assembly.window_reads(al, chrom_ids, [(l["chrom"], l["start"], l["end"]) for l in loci])
                names_p = [l["name"] for l in loci]; ctg = [l["contig"] for l in loci]
                telr_assembly.polish_consensus(eng, names_p[:8], ctg[:8], [w.astype(np.int32) for w in wr_p[:8]], presets=presets_arg, read_set=qs)      # sizes the scratch
                telr_assembly.polish_consensus(eng, names_p[:8], ctg[:8], [w.astype(np.int32) for w in wr_p[:8]], presets=presets_arg, read_set=qs, method="poa")      # (and the POA kernel's first launch: 4 s in one run of round 5)
                # three passes each, the median reported (as for the loci leg): a single pass of 0.1-0.2 s has been seen to take 3-4 s
                # once in a run (in round 5: the pile-up pass of the full default run, its phases below say where)
                def timed_polish(method):
                    runs = []
                    for _ in range(3):
                        ph = {}
                        sync(); t0p = time.time()
                        out = telr_assembly.polish_consensus(eng, names_p, ctg, [w.astype(np.int32) for w in wr_p], presets=presets_arg, read_set=qs, method=method, timings=ph)
                        sync(); runs.append((time.time() - t0p, ph))
                    order = sorted(range(3), key=lambda k: runs[k][0])
                    med = runs[order[1]]
                    return out, med[0], {k: round(v, 4) for k, v in med[1].items()}, [round(r[0], 4) for r in runs], {k: round(v, 4) for k, v in runs[order[2]][1].items()}
                pol, tp, php, each, slowest = timed_polish("pileup")
                polish = {"seconds": tp, "loci_per_s": len(loci) / tp, "seconds_of_each_pass": each, "phases_s": php, "phases_s_of_the_slowest_pass": slowest,
                          "contigs_changed": int(sum(1 for x, y in zip(pol, ctg) if x != y)),
                          "bases_before": int(sum(len(x) for x in ctg)), "bases_after": int(sum(len(x) for x in pol)),
                          "what": "one telr_map (-ax P -r2k, window reads of every locus against its draft contig) + one pile-up consensus pass over all contigs; median of three passes"}
                # the same hand-off with the window partial-order consensus (telr_poa_build)
                try:
                    pol2, tp2, ph2, each2, slowest2 = timed_polish("poa")
                    polish["poa"] = {"seconds": tp2, "loci_per_s": len(loci) / tp2, "seconds_of_each_pass": each2, "phases_s": ph2, "phases_s_of_the_slowest_pass": slowest2,
                                     "contigs_changed": int(sum(1 for x, y in zip(pol2, ctg) if x != y)),
                                     "bases_after": int(sum(len(x) for x in pol2)), "differs_from_pileup": int(sum(1 for x, y in zip(pol2, pol) if x != y)),
                                     "what": "one telr_map + one window partial-order consensus pass (200-base windows, one wave per window); median of three passes"}
                except Exception as e:
                    polish["poa"] = {"error": "%s: %s" % (type(e).__name__, e)}
            except Exception as e:
                polish = {"error": "%s: %s" % (type(e).__name__, e)}
            # the A/B DESIGN section 8 listed first: the call set of the SAME loci pass with the contigs polished on the device first
            # (TELR_assembly.py:185-262 runs wtpoa-cns there), pile-up against window POA against no polishing
            if isinstance(polish, dict) and "error" not in polish:
                ab = {}
                for mode in ("pileup", "poa"):
                    try:
                        sync(); t0p = time.time()
                        rows_m, _ = loci_pass(polish=mode)
                        sync(); tm_ = time.time() - t0p
                        g_, a_, w_ = tally(rows_m)
                        same = int(sum(1 for x, y in zip(np.sort(rows_m, order="locus_id"), np.sort(rows, order="locus_id"))
                                       if (int(x["locus_id"]), int(x["chrom_id"]), int(x["start"]), int(x["end"]), int(x["strand"]), int(x["type"]), [int(v) for v in x["family_id"]]) ==
                                          (int(y["locus_id"]), int(y["chrom_id"]), int(y["start"]), int(y["end"]), int(y["strand"]), int(y["type"]), [int(v) for v in y["family_id"]]))) if len(rows_m) == len(rows) else None
                        ab[mode] = {"recovered_exact_chrom_family_strand_pos20": g_, "of_those_af_within_0.15": a_, "rows_in_merged_table": len(rows_m), "not_recovered": w_,
                                    "rows_with_the_unpolished_call": same, "seconds_incl_polish": tm_}
                    except Exception as e:
                        ab[mode] = {"error": "%s: %s" % (type(e).__name__, e)}
                polish["call_set_ab"] = dict(ab, none={"recovered_exact_chrom_family_strand_pos20": good, "of_those_af_within_0.15": af_ok, "rows_in_merged_table": n_rows},
                                             what="the whole loci pass (S4-S7, depth, AF, liftover) on contigs polished first by the device consensus; `rows_with_the_unpolished_call` = rows whose "
                                                  "chromosome, coordinates, strand, type and families equal the unpolished pass's")
            # full-size parity of the two consensus kernels: the first --poa-parity loci, engine vs CPU oracle, contig string for contig string
            if a.poa_parity > 0 and isinstance(polish, dict) and "error" not in polish:
                try:
                    from oracle import binding as ob
                    npar = min(a.poa_parity, len(loci))
                    io_p, mo_p = preset("map-pb" if presets_arg == "pacbio" else "map-ont"); mo_p.bw = 2000
                    drafts = [c if isinstance(c, str) else bytes(c).decode() for c in ctg[:npar]]
                    qt_p = np.array([k for k in range(npar) for _ in wr_p[k]], np.int32)
                    ridx_p = np.concatenate([wr_p[k] for k in range(npar)]).astype(np.int32)
                    flat_p = [bytes(rbuf[roff[i]:roff[i] + rln[i]]).decode() for i in ridx_p]
                    ixp = eng.index(drafts, io_p); qsp = qs.subset(ridx_p)
                    rp = ixp.map_raw(qsp, mo_p, qtarget=qt_p)
                    resp = ixp.result_arrays(rp)
                    par = {"loci": npar, "reads": int(len(ridx_p)), "records": int(len(resp.alns)), "windows_of_200_bases": int(sum((len(d_) + 199) // 200 for d_ in drafts))}
                    for key, is_poa in (("poa", True), ("pileup", False)):
                        got = ixp.consensus(rp, qsp, min_depth=3, poa=is_poa)
                        t0o = time.time()
                        want = ob.consensus(resp.alns, resp.cigars, flat_p, drafts, min_depth=3, poa=is_poa)
                        par[key] = {"identical": got == want, "contigs_differing": int(sum(1 for x, y in zip(got, want) if x != y)), "contigs_changed_by_the_consensus": int(sum(1 for x, y in zip(got, drafts) if x != y)),
                                    "oracle_seconds": time.time() - t0o}
                    ixp.free_raw(rp); ixp.free(); qsp.free()
                    par["what"] = "telr_poa_build / telr_consensus_build on the polish alignments (-ax P -r2k, per-query targets) of the first loci with their real window reads, against the CPU oracle's tor_poa / pile-up on the same records: every contig string equal = every window equal"
                    polish["parity"] = par
                except Exception as e:
                    polish["parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
        wr_counts = [len(x) for x in telr_assembly.window_reads(al, chrom_ids, [(l["chrom"], l["start"], l["end"]) for l in loci])]
        import hashlib
        rs = np.sort(rows, order="locus_id")
        digest = hashlib.sha256(repr([(int(r["locus_id"]), int(r["status"]), int(r["chrom_id"]), int(r["start"]), int(r["end"]), int(r["strand"]), int(r["type"]), int(r["n_family"]),
                                       int(r["gap"]), int(r["tsd_len"]), [int(x) for x in r["family_id"]], [None if np.isnan(x) else float(x) for x in r["medians"]],
                                       None if np.isnan(r["af"]) else float(r["af"])) for r in rs]).encode()).hexdigest()
        loci_out = {"n": n_loci, "seconds": t_loci, "seconds_of_each_pass": t_passes, "device_free_GB_before_first_and_after_last_pass": [round(free_first / 1e9, 3), round(free_last / 1e9, 3)], "rows_in_merged_table": n_rows, "merged_table_sha256": digest, "recovered_exact_chrom_family_strand_pos20": good, "of_those_af_within_0.15": af_ok, "not_recovered": why,
                    "recovered_with_bw_long_0_at_S4_S6": alt_set,
                    "window_reads_per_locus_mean_this_rank": float(np.mean(wr_counts)) if wr_counts else 0.0, "polish_pileup": polish,
                    "collectives": "none" if world == 1 and not (a.force_exchange and dist is not None) else "all-to-all of the window reads as packed device words (counts + ONE int32 payload), ONE all-gather of the %d-byte locus rows" % shard.LOCUS_ROW.itemsize,
                    # where the last pass went on THIS rank (seconds): selecting + exchanging the window reads (pack_s = gather kernel and
                    # header upload, collective_s = the two all-to-alls and the header download), the bundle on the rank's loci, the one all-gather
                    "phase_s_last_pass_per_rank": phase_all,
                    "note": "host glue (Python) included; window reads = telr_assembly.window_reads on this run's stage-1 records; contigs / ALT sequences are "
                            "truth-derived stand-ins for wtdbg2 / Sniffles (absent on the box)"}
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    prim = al[(al["flags"] & 1) != 0]
    frac_mapped = len(np.unique(prim["qid"])) / max(1, len(D["reads"][2]))
    # roofline of the dominant kernel: k_dp_pk, the packed-int16 gap-fill DP (DP classes 10-17, gap fills by band width, in one cost-ordered launch
    # per range and lane).  Its launch durations are measured live with HIP events on the engine's own streams (telr_stage_ms: "k_dp_pk", summed
    # over the launches); its algorithmic bytes are counted per problem by the library: 2-bit query+target bases read once, 4 B per CIGAR run,
    # 32 B result.
    cls = cls_tot
    k_name = "k_dp_pk"
    rf0 = pk_roofline(T0)
    launches = max(1, launches)
    achieved, k_ms = rf0["achieved"], rf0["launch_ms"]
    k_ms_tot = stage_tot.get("k_dp_pk", 0.0)
    k_bytes_tot = float(cls[PK, 3].sum())
    # What this run did NOT measure itself -- HBM bytes and VALU issue of the kernel from the committed PMC passes of the same
    # command (rocprofv3 --pmc cannot run inside this process) -- is attached under `from_profile` with its provenance and only
    # when the profile was taken on this workload; everything else in `roofline` is measured live in this run.
    from_profile = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", PMC_PROFILE)))
        if pm.get("config") == a.config and not a.preset and not a.coverage and a.genome_scale == 1.0 and world == 1:
            from_profile = {"file": "profiles/" + PMC_PROFILE, "collected_at_commit": pm.get("head_sha"), "collected_on": pm.get("date"), "command": pm.get("command"),
                            "traffic_bytes_per_launch": pm["traffic_bytes_per_launch"], "traffic_over_algorithmic": pm.get("traffic_over_algorithmic"),
                            "first_pass_launches_in_pmc_run": pm.get("first_pass_launches_in_pmc_run"), "valu_issue": pm.get("valu_issue"), "how": pm.get("source")}
    except Exception:
        pass
    traffic = from_profile["traffic_bytes_per_launch"] if from_profile else None
    dp_ms = stage_tot.get("dp", 0.0) / a.steps
    ctr = {k: v / a.steps for k, v in ctr_tot.items()}
    # whole-path algorithmic bytes (SURVEY 8d formula) for reference
    path_bytes = (ctr["query_bases"] / 4.0 + 32.0 * ctr["minimizers"] + 16.0 * ctr["probes"] + 48.0 * ctr["anchors"]
                  + ctr["window_bases"] / 4.0 + 4.0 * ctr["cigar_ops"] + 64.0 * ctr["records"])
    gpu_ms = sum(v for k, v in stage_tot.items() if k not in ("select_host", "assemble_host", "index_build", "k_dp_reg", "map_wall", "k_traceback", "k_dp_pk")) / a.steps
    out = {
        "metric": "gbp_aligned_per_s", "value": value, "unit": "Gbp/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
        "dtype": "int16", "data": "synthetic",
        "value_incl_h2d": value_h2d, "value_streaming_incl_h2d": value_stream, "h2d_pack_upload_s": t_upload_max, "h2d_pack_upload_first_call_s": t_upload_first,
        "config": {"workload": "BASELINE %s: %s, preset %s, stage-1 reads->reference" % (cfg["label"], D["text"], pname + (" + experiment " + a.map_opt if a.map_opt else "")),
                   "reads_this_rank": int(len(D["reads"][2])), "read_bases_this_rank": n_bases, "read_bases_job": job_bases,
                   "parallelism": ("one fixed read set dealt to %d ranks in blocks by cumulative bases" % world if a.scaling == "strong" else "every rank maps its own read set (x%d)" % world)
                                  + "; index replicated (built by every rank, no broadcast); no collective on the stage-1 data path",
                   "ranges": "one telr_map call per step; a read set of up to 1.6 Gbp is one range, a larger one streams through in an even number of equal ranges of at most 1.4 Gbp, two in flight (whichever slot is free takes the next), each also bounded by 1.6 G anchors at the density the index has shown (include/telr_hip.h: telr_map)",
                   "streaming": "telr_map returns with the records; the CIGAR DMA of step k overlaps step k+1 (all complete inside the timed region)"},
        "per_rank_ms_per_step": per_rank_ms, "rccl_world_size": world if dist is not None else 0,
        "roofline": {"bound": "hbm", "kernel": k_name, "dp_classes": PK, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                     "traffic": traffic, "from_profile": from_profile, "launch_ms": k_ms, "launches_per_step": launches / a.steps,
                     "algorithmic_bytes_per_launch": k_bytes_tot / launches,
                     "problems_per_launch": float(cls[PK, 0].sum()) / launches, "cells_per_launch": float(cls[PK, 1].sum()) / launches,
                     "gcups": float(cls[PK, 1].sum()) / (k_ms_tot * 1e-3) / 1e9 if k_ms_tot > 0 else None,
                     "note": "the contract's HBM roofline of the dominant kernel; the kernel's binding resource is VALU issue (see roofline_valu_issue). With two ranges in flight a launch shares the device with the other range's kernels, so launch_ms is its stretched duration; all DP kernels together: %.1f ms per step, %.0f GCUPS"
                             % (dp_ms, ctr["dp_cells"] / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0)},
        # the same kernel against the resource that binds it: wave-instructions issued per SIMD-cycle against the issue interval
        # of its instruction mix measured on this part (tools/ubench/valu_rate.hip).  Counter-derived: present only with a profile.
        "roofline_valu_issue": None if not (from_profile and from_profile.get("valu_issue")) else {
            "bound": "valu_issue", "kernel": k_name, "achieved": from_profile["valu_issue"]["cycles_per_wave_instruction_per_simd"],
            "peak": from_profile["valu_issue"]["measured_issue_interval_cycles"], "unit": "cycles per wave-instruction per SIMD (lower is better)",
            "frac": from_profile["valu_issue"]["valu_issue_frac"], "from_profile": from_profile["file"]},
        "stage_ms_per_step": {k: v / a.steps for k, v in stage_tot.items()},
        "dp_classes": {str(c): [int(x) // a.steps for x in cls[c]] for c in range(cls.shape[0]) if cls[c, 0]},
        "path_algorithmic_GBps": path_bytes / (gpu_ms * 1e-3) / 1e9 if gpu_ms > 0 else None,
        "frac_reads_mapped": frac_mapped, "index_build_s": t_index, "datagen_s": t_gen, "device": eng.device_name(),
        "counters": ctr, "dp_retries": retries // a.steps,
    }
    if bam_out is not None:
        out["stage1_to_sorted_bam"] = bam_out
    if shard_out is not None:
        # end to end (round 6): map step + the rank's local BAM phases + the loci leg, rank 0's share of an N-rank run against the one-GPU run of the
        # same three -- still an estimate from ONE device (no exchange over xGMI, no skew between ranks)
        try:
            ls = locals().get("loci_shard_s") or {}
            if "ranks" in shard_out and bam_out is not None and bam_out.get("bam_seconds") is not None and loci_out is not None and "error" not in ls:
                ms_step = dt / a.steps * 1e3
                t1 = ms_step / 1e3 + bam_out["bam_seconds"] + loci_out["seconds"]
                e2e = {"one_gpu_seconds": {"map_step": ms_step / 1e3, "bam_local": bam_out["bam_seconds"], "loci_pass": loci_out["seconds"], "sum": t1}, "ranks": {}}
                for nrk, rv in shard_out["ranks"].items():
                    if rv.get("bam_seconds_local") is None or nrk not in ls:
                        continue
                    tn = rv["ms_per_step"] / 1e3 + rv["bam_seconds_local"] + ls[nrk]["seconds"]
                    e2e["ranks"][nrk] = {"map_step": rv["ms_per_step"] / 1e3, "bam_local": rv["bam_seconds_local"], "loci_pass": ls[nrk]["seconds"], "loci": ls[nrk]["loci"], "sum": tn,
                                         "speedup_vs_one_gpu": t1 / tn, "efficiency": t1 / tn / int(nrk)}
                e2e["what"] = "rank 0's share of an N-rank run (reads by cumulative bases, loci by LPT) through the map step, the device BAM writer and the loci pass, alone on this GPU, against the one-GPU run of the same three: what the local work allows, before the exchanges (records + packed reads of the job BAM, window reads, ONE all-gather of 104-byte rows) and rank skew"
                shard_out["end_to_end"] = e2e
        except Exception as e:
            shard_out["end_to_end"] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["expected_strong_scaling"] = shard_out
    if a.loci and locals().get("flank_parity") is not None:
        out["flank_parity_asm10"] = flank_parity
    if files_out is not None:
        out["stage1_from_files"] = files_out
    if loci_out is not None:
        out["te_loci_per_s"] = loci_out["n"] / loci_out["seconds"]
        out["te_loci"] = loci_out
    # ---- the reference's own CPU path (ngmlr / minimap2 / samtools / bedtools), when this box has it on PATH: timed with the
    # reference's argv shapes S1 / S2 / S7 and cross-checked record by record against this engine (tools/crosscheck_upstream.py;
    # SURVEY 8(d)).  Absent tools are reported as absent.
    if not a.no_upstream_check:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import crosscheck_upstream as xc

            def xc_sample():
                rng = np.random.default_rng(a.cpu_sample_seed)
                rb, ro, rl = D["reads"]
                pick = np.sort(rng.choice(len(rl), size=min(len(rl), a.upstream_sample_reads), replace=False))
                fl_n, fl_s = [], []
                for l in D["loci"][:100]:
                    c = l["contig"] if isinstance(l["contig"], (bytes, bytearray)) else l["contig"].encode()
                    fl_n += [l["name"] + "_5p", l["name"] + "_3p"]; fl_s += [c[:500], c[-500:]]
                return dict(ref_names=D["names"], ref_seqs=ref_strs, read_names=["read%d" % D["read_gid"][i] for i in pick],
                            read_seqs=[bytes(rb[ro[i]:ro[i] + rl[i]]) for i in pick], read_bases=int(rl[pick].sum()), flank_names=fl_n, flank_seqs=fl_s,
                            text="%d reads drawn at random (seed %d) from this run's read set + the first and last 500 bases of the first 100 locus contigs, against the full reference" % (len(pick), a.cpu_sample_seed))

            def xc_bam(ref_fa, reads_fa, bam_path):
                from telr_amd import telr_alignment
                telr_alignment.alignment(bam_path, reads_fa, ref_fa, os.path.dirname(bam_path), "xcheck", 1, "minimap2", "ont", engine=eng)
                return None
            out["reference_cpu_path"] = xc.reference_cpu_path(ours=xc.default_ours(eng), sample=xc_sample, presets="ont" if cfg["err"][1] < 0.05 else "pacbio", bam_writer=xc_bam)
        except Exception as e:
            out["reference_cpu_path"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if default_aligner is not None:
        out["value_reference_default_aligner"] = default_aligner
        if "error" not in default_aligner and a.default_aligner_parity_reads > 0 and not a.no_cpu_baseline:
            # the same leg's parity evidence and CPU figure: the oracle with the same preset on a random sample of the same reads
            cb = cpu_baseline(ref_strs, D["reads"], io_ng, mo_ng, a.default_aligner_parity_reads, gpu_index=ix_ng, seed=a.cpu_sample_seed + 1)
            default_aligner["parity"] = cb.pop("parity", None)
            default_aligner["cpu_baseline"] = cb
    if not a.no_cpu_baseline:
        ns = a.cpu_sample_reads or max(8, int(3.0e7 * usable_cpus() / max(1.0, n_bases / len(D["reads"][2]))))   # ~15-20 s of CPU work
        out["cpu_baseline"] = cpu_baseline(ref_strs, D["reads"], io, mo, ns, gpu_index=ix, seed=a.cpu_sample_seed)
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"] if out["cpu_baseline"]["value"] > 0 else None
    json_out.write(json.dumps(out) + "\n"); json_out.flush()
    sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
