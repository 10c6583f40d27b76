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
           "--reads", "1500", "--read-bases", "60000000", "--insertions", "30", "--cpu-sample-reads", "40", "--default-aligner-parity-reads", "40"]
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
    assert (r["bound"], r["peak"]) in {("hbm", 8000.0), ("mfma", 2500.0)} and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and 0 < c["value"] < d["value"]
    assert c["parity"]["identical"] is True and c["parity"]["reads"] == 40 and c["parity"]["records_engine"] == c["parity"]["records_oracle"] > 0
    # the reference's DEFAULT aligner (`--aligner nglmr`) under the same clock: same steps, own roofline, own parity sample
    g = d["value_reference_default_aligner"]
    assert "error" not in g, g
    assert g["preset"] == "ngmlr-ont" and g["steps"] == 2 and g["warmup"] == 1 and 0 < g["value"] and g["unit"] == "Gbp/s"
    assert abs(g["roofline"]["frac"] - g["roofline"]["achieved"] / 8000.0) < 1e-9 and g["roofline"]["bound"] == "hbm"
    assert g["parity"]["identical"] is True and g["parity"]["reads"] == 40 and g["parity"]["records_engine"] == g["parity"]["records_oracle"] > 0
    assert g["cpu_baseline"]["kind"] == "port" and 0 < g["cpu_baseline"]["value"] < g["value"]
    # the upstream tools are not on this box: the cross-check leg says so instead of skipping silently
    x = d["reference_cpu_path"]
    assert x["looked_for"] == ["minimap2", "ngmlr", "samtools", "bedtools"] and (x["available"] is False or "shapes" in x)
    b = d["stage1_to_sorted_bam"]                                        # the default run carries the stage-1 hand-off leg
    assert "error" not in b and b["writer"] == "device" and 0 < b["gbp_per_s_incl_bam"] < d["value"] and b["bam_bytes"] > 1000000
    assert d["value_incl_h2d"] <= d["value"] and d["te_loci"]["n"] == 30
    # measured with every step's reads packed and uploaded underneath the previous step; two steps of 25 ms each: a relation between two
    # noisy numbers, only there to catch a wrong unit
    assert 0 < d["value_streaming_incl_h2d"] <= 2.0 * d["value"]


@pytest.mark.gpu
def test_two_ranks_merge_to_the_one_rank_locus_table():
    """The N > 1 code path on the real engine (two ranks on device 0, gloo): reads dealt to the ranks, every rank maps its
    share, the window reads of a locus travel to the locus' owner (all-to-all), the per-locus rows are merged by ONE
    all-gather -- and the merged table is the table one rank computes alone."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--config", "c1", "--genome-len", "3000000", "--reads", "1500",
            "--read-bases", "60000000", "--insertions", "30", "--no-cpu-baseline", "--no-stream-leg", "--bam-sha"]
    out = []
    for extra in (["--gpus", "1"], ["--gpus", "2", "--one-gpu", "--backend", "gloo"]):
        env = dict(os.environ)
        if "2" in extra:
            env["TELR_KEEP_BAM"] = "1"                 # the job's file is validated below
        p = subprocess.run(base + extra, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
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
    # stage 1 hands Sniffles ONE sorted BAM: both ranks write their coordinate slice of the one file (shard.write_job_bam); its INFLATED
    # stream is, byte for byte, the stream of the file one rank writes alone (the BGZF block boundaries differ: two coders)
    jb = two["stage1_to_sorted_bam"]["job_bam"]
    assert "error" not in jb, jb
    assert one["stage1_to_sorted_bam"]["job_bam"] is None
    assert jb["inflated_sha256"] == one["stage1_to_sorted_bam"]["inflated_sha256"] is not None
    assert jb["reads"] == one["config"]["reads_this_rank"] > two["config"]["reads_this_rank"] and jb["records"] > 0
    ph = jb["phase_s_per_rank"]
    assert len(ph) == 2 and all(k in ph[r] for r in (0, 1) for k in ("partition_s", "collective_s", "code_slice_s", "write_slice_s", "index_s", "slice_bytes"))
    assert ph[0]["slice_bytes"] > 0 and ph[1]["slice_bytes"] > 0 and ph[0]["slice_bytes"] + ph[1]["slice_bytes"] + 28 == jb["bam_bytes"]
    assert ph[0]["slice_unmapped_reads"] == 0                                 # reads without a record belong to the last slice
    # the merged index: every chunk and every linear-index entry of the .bai points at a record start of the two-coder file
    v = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "validate_bam.py"), jb["path"], str(jb["records"] + ph[1]["slice_unmapped_reads"])],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    for f in (jb["path"], jb["path"] + ".bai"):
        if os.path.exists(f):
            os.unlink(f)
    assert v.returncode == 0, v.stderr.decode()[-2000:]
    vj = json.loads(v.stdout.decode().strip().splitlines()[-1])
    assert vj["records"] == jb["records"] + ph[1]["slice_unmapped_reads"] and vj["bai_chunks"] > 0 and vj["bai_linear_entries"] > 0


@pytest.mark.gpu
def test_collectives_of_the_n_rank_path_run_through_rccl_on_device_tensors():
    """torch.distributed.run with ONE rank and --force-exchange: the all-to-all / all-gather of the loci leg and the stage-1
    gather of the job BAM go through RCCL (backend nccl) on device tensors, and give what the collective-free path gives."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    args = ["--gpus", "1", "--steps", "1", "--warmup", "0", "--config", "c1", "--genome-len", "3000000", "--reads", "1500", "--read-bases", "60000000",
            "--insertions", "30", "--no-cpu-baseline", "--no-stream-leg", "--bam-sha"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    plain = json.loads([l for l in p.stdout.decode().splitlines() if l.strip()][-1])
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py")] + args + ["--force-exchange"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    forced = json.loads([l for l in p.stdout.decode().splitlines() if l.strip().startswith("{")][-1])
    assert forced["rccl_world_size"] == 1 and plain["rccl_world_size"] == 0
    assert "all-to-all" in forced["te_loci"]["collectives"]
    assert forced["te_loci"]["merged_table_sha256"] == plain["te_loci"]["merged_table_sha256"]
    jb = forced["stage1_to_sorted_bam"]["job_bam"]
    assert "error" not in jb and "nccl" in jb["what"], jb
    assert jb["inflated_sha256"] == plain["stage1_to_sorted_bam"]["inflated_sha256"] == forced["stage1_to_sorted_bam"]["inflated_sha256"] is not None
    # round 6: the record exchange's payload never leaves the device under RCCL -- packed on it, sent from it, received on it --
    # and the all-to-all really carried bytes (at world size 1: all of them, to the rank itself)
    ph = jb["phase_s_per_rank"][0]
    assert ph["wire"].startswith("cuda") and ph["payload_sent_from"].startswith("cuda") and ph["payload_received_on"].startswith("cuda"), ph
    assert ph["bytes_to_self"] > 1000000 and ph["bytes_sent"] == 0, ph
