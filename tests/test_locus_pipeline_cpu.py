"""The host pipeline of the per-locus bundle, driven by the CPU oracle (no GPU): spiked insertions are
recovered with the right coordinate, family, strand and a sensible allele frequency."""
import sys
import os

sys.path.insert(0, os.path.dirname(__file__))

from telr_amd import locus_pipeline
from telr_amd.presets import preset


def test_locus_bundle_on_oracle_recovers_truth():
    from oracle_backend import OracleBackend
    from locus_data import make_loci
    from test_gpu_locus import check_truth
    ref, lib_names, lib, loci, truth = make_loci(n_ins=4, reads_per_locus=20)
    be = OracleBackend()
    io, _ = preset("asm10")
    res = locus_pipeline.run_loci(be, be.index([ref], io), ["chr2L"], lambda ch: ref, loci, lib_names, lib, presets="ont")
    check_truth(res, loci, truth)
    assert len(res["annotation"]) >= 3
