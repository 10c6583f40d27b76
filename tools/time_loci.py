"""Where the per-locus bundle (bench.py's TE-loci/s leg) spends its time: cProfile of run_loci on the bench dataset."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from telr_amd import synth, locus_pipeline
from telr_amd.aligner import Engine
from telr_amd.presets import preset

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
d = synth.make_stage1_dataset(genome_len=23513712, n_reads=10000, seed=20261002, n_ins=200, read_seed=20261002 + 1000) if False else \
    synth.make_stage1_dataset(genome_len=6_000_000, n_reads=2500, seed=20261002, n_ins=n)
ref_str = bytes(d["ref"]).decode()
eng = Engine(0)
loci = synth.make_loci_from_dataset(d, min(n, len(d["insertions"])))
io10, _ = preset("asm10")
ix10 = eng.index([ref_str], io10)
lib_names = ["fam%d" % i for i in range(len(d["library"]))]
lib = [bytes(x).decode() for x in d["library"]]
locus_pipeline.run_loci(eng, ix10, ["chr2L"], lambda ch: ref_str, loci[:8], lib_names, lib)
t0 = time.time()
pr = cProfile.Profile(); pr.enable()
locus_pipeline.run_loci(eng, ix10, ["chr2L"], lambda ch: ref_str, loci, lib_names, lib)
pr.disable()
print("loci", len(loci), "seconds", time.time() - t0)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
