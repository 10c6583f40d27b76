import cProfile, pstats, sys, os, time
sys.path.insert(0, os.getcwd())
from telr_amd import synth, locus_pipeline
from telr_amd.aligner import Engine
from telr_amd.presets import preset
d = synth.make_stage1_dataset(genome_len=23513712, n_reads=10000, total_bases=470_000_000, seed=20261002, n_ins=200, read_seed=20261002 + 1000)
ref_str = bytes(d["ref"]).decode()
eng = Engine(0)
qs = eng.seqset(d["reads"])
loci = synth.make_loci_from_dataset(d, 200)
io10, _ = preset("asm10")
ix10 = eng.index([ref_str], io10)
lib_names = ["fam%d" % i for i in range(len(d["library"]))]
lib = [bytes(x).decode() for x in d["library"]]
locus_pipeline.run_loci(eng, ix10, ["chr2L"], lambda ch: ref_str, loci[:8], lib_names, lib, read_set=qs)
locus_pipeline.run_loci(eng, ix10, ["chr2L"], lambda ch: ref_str, loci, lib_names, lib, read_set=qs)
t0 = time.time()
pr = cProfile.Profile(); pr.enable()
locus_pipeline.run_loci(eng, ix10, ["chr2L"], lambda ch: ref_str, loci, lib_names, lib, read_set=qs)
pr.disable()
print("loci", len(loci), "seconds", time.time() - t0)
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
print("stages of the last engine call (AF realignment):", {k: round(v, 2) for k, v in eng.stage_ms().items() if v > 0.01})
print("counters:", eng.counters())
