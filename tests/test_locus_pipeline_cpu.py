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


def test_bundle_to_output_files(tmp_path):
    """run_loci -> write_outputs: the six result files of a TELR run come out of the in-memory results, one VCF/BED row
    and one TE sequence per non-reference insertion, coordinates as in the liftover report."""
    from oracle_backend import OracleBackend
    from locus_data import make_loci
    ref, lib_names, lib, loci, truth = make_loci(n_ins=4, reads_per_locus=20)
    be = OracleBackend()
    io, _ = preset("asm10")
    res = locus_pipeline.run_loci(be, be.index([ref], io), ["chr2L"], lambda ch: ref, loci, lib_names, lib, presets="ont")
    ref_fa = tmp_path / "ref.fa"
    ref_fa.write_text(">chr2L\n" + "\n".join(ref[i:i + 60] for i in range(0, len(ref), 60)) + "\n")
    final, expanded = locus_pipeline.write_outputs(res, loci, str(tmp_path), "s", str(ref_fa), today="DATE")
    nonref = [r for r in res["liftover"] if r["report"]["type"] == "non-reference"]
    assert len(final) == len(nonref) >= 3
    vcf = (tmp_path / "s.telr.vcf").read_text().splitlines()
    body = [l for l in vcf if not l.startswith("#")]
    assert "##contig=<ID=chr2L,length=%d>" % len(ref) in vcf and len(body) == len(final)
    bed = (tmp_path / "s.telr.bed").read_text().splitlines()
    for row, rep in zip(bed, final):
        f = row.split("\t")
        assert f[0] == "chr2L" and int(f[1]) == rep["start"] and int(f[2]) == rep["end"] and f[3] == rep["family"] and f[5] == rep["strand"]
    te = (tmp_path / "s.telr.te.fasta").read_text().count(">")
    assert te == len(final) and (tmp_path / "s.telr.json").exists() and (tmp_path / "s.telr.expanded.json").exists()
    assert all(r["te_length"] == len(r["te_sequence"]) for r in expanded)


def test_chromosome_name_with_underscores():
    """a scaffold-style name (chrUn_CP007071v1): loci keep their rows through run_loci_distributed (the report -> locus key
    is everything before the last two '_' fields), the 5' chromosome filter of the liftover sees the full name, and loci
    that carry read_idx / read_bases instead of read sequences can be dealt by cost"""
    from oracle_backend import OracleBackend
    from locus_data import make_loci
    ref, lib_names, lib, loci, truth = make_loci(n_ins=3, reads_per_locus=16)
    chrom = "chrUn_CP007071v1"
    for l in loci:
        l["name"] = l["name"].replace("chr2L", chrom)
    be = OracleBackend()
    io, _ = preset("asm10")
    rows, res = locus_pipeline.run_loci_distributed(be, be.index([ref], io), [chrom], lambda ch: ref, loci, lib_names, lib, presets="ont")
    assert len(rows) == len(loci) and set(rows["locus_id"].tolist()) == set(range(len(loci)))
    ok = sum(1 for r in rows if r["type"] == 1 and r["chrom_id"] == 0 and abs(int(r["start"]) - truth[int(r["locus_id"])]["pos"]) <= 20)
    assert ok >= len(loci) - 1
    assert all(r["report"]["chrom"] in (chrom, None) for r in res["liftover"])
    assert locus_pipeline.locus_cost({"contig": "A" * 10, "alt": None, "read_idx": [1, 2]}) == 10
