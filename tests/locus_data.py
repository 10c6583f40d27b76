"""Small seeded locus data set shared by the CPU and GPU pipeline tests."""
import numpy as np

from telr_amd import synth


def S(a):
    return bytes(a).decode()


def make_loci(seed=5, genome_len=400000, n_ins=8, n_fam=6, reads_per_locus=30):
    rng = np.random.default_rng(seed)
    ref = synth.random_seq(rng, genome_len)
    lib = [synth.random_seq(rng, int(L), gc=0.45) for L in rng.integers(700, 4500, size=n_fam)]
    lib_names = ["fam%d" % i for i in range(n_fam)]
    # a few diverged reference copies
    for _ in range(12):
        f = lib[int(rng.integers(0, n_fam))]
        cp = synth.mutate(rng, f, 0.08, 0.0, 0.0)
        p = int(rng.integers(0, genome_len - len(cp)))
        ref[p:p + len(cp)] = cp
    sites = np.sort(rng.choice(np.arange(20000, genome_len - 20000, 15000), size=n_ins, replace=False)) + rng.integers(0, 3000, size=n_ins)
    loci, truth = [], []
    for p in sites:
        p = int(p); fam = int(rng.integers(0, n_fam)); strand = int(rng.integers(0, 2)); tsd = int(rng.integers(4, 9))
        af = float(rng.choice([0.5, 1.0]))
        te = lib[fam] if not strand else synth.revcomp_arr(lib[fam])
        hap = np.concatenate([ref[p - 12000:p + tsd], te, ref[p:p + 12000]])          # insertion allele with TSD
        refhap = ref[p - 12000:p + 12000]
        ins_at = 12000 + tsd
        lo = int(rng.integers(8000, 10000)); hi = int(rng.integers(8000, 10000))
        contig = synth.mutate(rng, hap[ins_at - lo:ins_at + len(te) + hi], 0.003, 0.001, 0.001)
        alt = synth.mutate(rng, te, 0.03, 0.01, 0.01)
        reads = []
        for _ in range(reads_per_locus):
            from_ins = rng.random() < af
            h = hap if from_ins else refhap
            centre = ins_at + (len(te) // 2 if from_ins else 0)
            L = int(rng.integers(6000, 14000))
            s = max(0, min(len(h) - L, centre - int(rng.integers(1500, L - 1500))))
            r = h[s:s + L]
            if rng.integers(0, 2):
                r = synth.revcomp_arr(r)
            reads.append(S(synth.mutate(rng, r, 0.03, 0.015, 0.015)))
        loci.append({"name": "chr2L_%d_%d" % (p, p + 1), "contig": S(contig), "alt": S(alt), "reads": reads})
        truth.append({"pos": p, "family": lib_names[fam], "strand": "+-"[strand], "tsd": tsd, "af": af, "te_len": len(te)})
    return S(ref), lib_names, [S(x) for x in lib], loci, truth
