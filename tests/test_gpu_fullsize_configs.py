"""BASELINE configs[3] and configs[4] at FULL size under the driver's eyes: one step of the default
bench on the configuration's whole read set, and the oracle's records for a random sample of the reads -- mapped against the
same full-size index -- must equal the engine's, record for record and CIGAR for CIGAR.  And the per-locus call sites at
configs[2]'s full size: S7 (flanks, asm10 -N 10, against the 137.6-Mb reference), S6 (window reads against the forward /
reverse-complement contig of their locus) and the whole bundle, engine run == oracle run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-stream-leg", "--no-shard-leg", "--bam-leg", "none"] + list(args)
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines[:3]
    return json.loads(lines[0])


@pytest.mark.parametrize("cfg,preset,min_bases", [("c3", "ngmlr-pacbio", 3.5e9), ("c4", "map-ont", 2.0e9), ("c2", "ngmlr-ont", 3.5e9)])
def test_full_size_configuration_equals_the_oracle_on_a_random_sample(cfg, preset, min_bases):
    """(the third case: the reference's default aligner on ONT reads -- `ngmlr -x ont`, NGMLR's convex gap cost at scale 10 -- on the
    configs[2] read set)"""
    d = _bench("--config", cfg, "--loci", "0", "--cpu-sample-reads", "4000", *(["--preset", preset] if cfg == "c2" else []))
    assert ("preset " + preset) in d["config"]["workload"] and d["config"]["read_bases_this_rank"] >= min_bases, d["config"]
    par = d["cpu_baseline"]["parity"]
    assert par["reads"] == 4000 and par["identical"] is True and par["reads_differing"] == 0 and par["records_engine"] == par["records_oracle"] >= 3000, par
    assert d["frac_reads_mapped"] > 0.75 and d["value"] > 1.0


@pytest.mark.parametrize("preset", ["ngmlr-ont"])          # (`map-ont` on the hard genome: the c2r bench line carries its own 54,444-read parity sample, profiles/r06_bench_c2r.json)
def test_hard_genome_at_configs2_size_equals_the_oracle_on_a_random_sample(preset):
    """round 6: `--config c2r` = configs[2] on the HARD genome (4 % tandem arrays, microsatellites, low-complexity
    stretches, segmental duplications, satellite blocks next to 15 % of the insertions; reads with error bursts), both stage-1
    aligner (the reference's default): one step on the whole read set, 2,000 sampled reads record for record against the oracle on the same full-size index;
    the over-size path is in use (reads inside arrays hold more anchors than one workgroup sorts in LDS)"""
    d = _bench("--config", "c2r", "--loci", "0", "--cpu-sample-reads", "2000", "--no-default-aligner-leg", "--preset", preset)
    assert "hard genome" in d["config"]["workload"] and ("preset " + preset) in d["config"]["workload"] and d["config"]["read_bases_this_rank"] >= 3.5e9, d["config"]
    par = d["cpu_baseline"]["parity"]
    assert par["reads"] == 2000 and par["identical"] is True and par["reads_differing"] == 0 and par["records_engine"] == par["records_oracle"] >= 1600, par
    assert d["counters"]["over_queries"] > 0 and d["frac_reads_mapped"] > 0.75


def test_call_sites_s6_s7_and_the_bundle_at_configs2_size():
    d = _bench("--config", "c2", "--loci", "100", "--flank-parity", "--no-cpu-baseline", "--no-polish-leg")
    fp = d["flank_parity_asm10"]
    assert fp["parity"]["identical"] is True and fp["parity"]["reads"] == 200 and fp["parity"]["records_engine"] >= 200, fp["parity"]      # S7: 2 flanks per locus
    assert fp["s6"]["identical"] is True and fp["s6"]["loci"] == 100 and fp["s6"]["records_engine"] == fp["s6"]["records_oracle"] > 5000, fp["s6"]
    assert fp["bundle"]["identical"] is True and fp["bundle"]["loci"] == 60 and fp["bundle"]["liftover_reports"] == 60, fp["bundle"]


def test_polish_hand_off_at_configs2_size_poa_and_pile_up_equal_the_oracle():
    """H3 (TELR_assembly.py:226-247) at full size: the polish alignments of 100 configs[2] loci with their real window reads (~4,000
    reads, ~12,600 windows of 200 bases); telr_poa_build == tor_poa and telr_consensus_build == the oracle's pile-up for every contig,
    i.e. for every window; and the call-set A/B of the three polish modes is in the line"""
    d = _bench("--config", "c2", "--loci", "100", "--poa-parity", "100", "--no-cpu-baseline", "--no-default-aligner-leg")
    pol = d["te_loci"]["polish_pileup"]
    assert "error" not in pol, pol
    par = pol["parity"]
    assert "error" not in par, par
    assert par["loci"] == 100 and par["reads"] > 2000 and par["records"] > 2000 and par["windows_of_200_bases"] > 10000, par
    for k in ("poa", "pileup"):
        assert par[k]["identical"] is True and par[k]["contigs_differing"] == 0 and par[k]["contigs_changed_by_the_consensus"] > 50, (k, par[k])
    ab = pol["call_set_ab"]
    for k in ("pileup", "poa"):
        assert "error" not in ab[k] and ab[k]["rows_in_merged_table"] == ab["none"]["rows_in_merged_table"] == 100, ab
