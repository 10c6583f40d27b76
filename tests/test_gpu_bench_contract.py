"""The driver's contract on `bench.py` (one JSON line on stdout, the keys it reads, the two added objects), checked on a
reduced configs[1]-shaped workload so that the whole run takes seconds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--config", "c1", "--genome-len", "3000000",
           "--reads", "1500", "--read-bases", "60000000", "--insertions", "30", "--cpu-sample-reads", "40"]
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines[:5]                       # nothing but the line (library banners go to stderr)
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "value_incl_h2d", "te_loci_per_s"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "Gbp/s" and d["value"] > 0.05 and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    whole = d["config"]["read_bases_this_rank"] / (d["ms_per_step"] * 1e-3) / 1e9          # every base of every read, per second
    assert 0.8 * whole <= d["value"] <= 1.001 * whole                                          # value counts the bases of reads with a primary record
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and 0 < c["value"] < d["value"]
    assert c["parity"]["identical"] is True and c["parity"]["reads"] == 40 and c["parity"]["records_engine"] == c["parity"]["records_oracle"] > 0
    b = d["stage1_to_sorted_bam"]                                        # the default run carries the stage-1 hand-off leg
    assert "error" not in b and b["writer"] == "device" and 0 < b["gbp_per_s_incl_bam"] < d["value"] and b["bam_bytes"] > 1000000
    assert d["value_incl_h2d"] <= d["value"] and d["te_loci"]["n"] == 30
    assert 0 < d["value_streaming_incl_h2d"] <= 1.3 * d["value"]          # measured with every step's reads packed and uploaded underneath the previous step


@pytest.mark.gpu
def test_two_ranks_merge_to_the_one_rank_locus_table():
    """The N > 1 code path on the real engine (two ranks on device 0, gloo): reads dealt to the ranks, every rank maps its
    share, the window reads of a locus travel to the locus' owner (all-to-all), the per-locus rows are merged by ONE
    all-gather -- and the merged table is the table one rank computes alone."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--config", "c1", "--genome-len", "3000000", "--reads", "1500",
            "--read-bases", "60000000", "--insertions", "30", "--no-cpu-baseline", "--no-stream-leg"]
    out = []
    for extra in (["--gpus", "1"], ["--gpus", "2", "--one-gpu", "--backend", "gloo"]):
        p = subprocess.run(base + extra, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert p.returncode == 0, p.stderr.decode()[-3000:]
        lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
        assert len(lines) == 1, lines[:5]
        out.append(json.loads(lines[0]))
    one, two = out
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert len(two["per_rank_ms_per_step"]) == 2 and all(x > 0 for x in two["per_rank_ms_per_step"])
    assert abs(two["config"]["read_bases_job"] - one["config"]["read_bases_job"]) < 1 and two["config"]["read_bases_this_rank"] < 0.6 * one["config"]["read_bases_job"]
    assert one["te_loci"]["collectives"] == "none"
    assert "all-to-all" in two["te_loci"]["collectives"] and "ONE all-gather" in two["te_loci"]["collectives"]
    assert two["te_loci"]["rows_in_merged_table"] == one["te_loci"]["rows_in_merged_table"] > 0
    assert two["te_loci"]["merged_table_sha256"] == one["te_loci"]["merged_table_sha256"]
    assert two["te_loci"]["recovered_exact_chrom_family_strand_pos20"] == one["te_loci"]["recovered_exact_chrom_family_strand_pos20"] >= 25
