"""Stage 1 from files, over and over: telr_alignment.alignment() on a configs[1]-size read set N times in one process.
Every run must write the same BAM and .bai (SHA-256), the device must not lose memory from run to run, and the wall clock
must not creep (prepared sink, background releases, the device copy of the CIGARs, pooled buffers: all of it is reused)."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from telr_amd import synth, telr_alignment
from telr_amd.aligner import Engine

n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
d = synth.make_stage1_dataset(genome_len=23513712, n_reads=10000, total_bases=470_000_000, seed=20261002, n_ins=200, read_seed=20261002 + 1000)
tmp = "/dev/shm" if os.access("/dev/shm", os.W_OK) else "/tmp"
rf, qf, bam = (os.path.join(tmp, "telr_soak_" + x) for x in ("ref.fa", "reads.fa", "out.bam"))
with open(rf, "wb") as fh:
    fh.write(b">chr2L\n" + bytes(d["ref"]) + b"\n")
buf, off, ln = d["reads"]
with open(qf, "wb") as fh:
    for i in range(len(ln)):
        fh.write(b">read%d\n" % i + bytes(buf[off[i]:off[i] + ln[i]]) + b"\n")
eng = Engine(0)


def sha(p):
    h = hashlib.sha256()
    with open(p, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


import psutil
proc = psutil.Process()
times, frees, shas, rss = [], [], set(), []
for i in range(n_runs):
    for f in (bam, bam + ".bai"):
        if os.path.exists(f):
            os.unlink(f)
    eng.L.telr_bam_release_wait()
    t0 = time.time()
    telr_alignment.alignment(bam, qf, rf, tmp, "soak", 1, "minimap2" if i % 2 == 0 else "nglmr", "ont", engine=eng)
    times.append(time.time() - t0)
    shas.add((i % 2, sha(bam), sha(bam + ".bai")))
    frees.append(eng.mem_info()[0])
    time.sleep(0.3)                      # the background release of the run (read set, index, file mappings)
    rss.append(proc.memory_info().rss)
for f in (rf, qf, bam, bam + ".bai"):
    if os.path.exists(f):
        os.unlink(f)
out = {"runs": n_runs, "distinct_outputs_per_method": {m: len([1 for s in shas if s[0] == m]) for m in (0, 1)}, "seconds_first": times[0], "seconds_median": float(np.median(times[2:])),
       "seconds_max_after_warmup": max(times[2:]), "free_GB_after_run_2": frees[2] / 1e9, "free_GB_after_last": frees[-1] / 1e9,
       "host_rss_GB_after_run_4": rss[4] / 1e9 if len(rss) > 4 else None, "host_rss_GB_after_last": rss[-1] / 1e9}
print(json.dumps(out))
assert all(v == 1 for v in out["distinct_outputs_per_method"].values()), "outputs differ between runs"
assert frees[-1] >= frees[3] - (1 << 30), "device memory shrinks from run to run"
assert len(rss) <= 6 or rss[-1] <= rss[4] + (1 << 30), "host memory grows from run to run"
