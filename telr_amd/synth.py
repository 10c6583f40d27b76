"""Seeded synthetic genomes / TE libraries / long reads (SURVEY.md 8d recipe, numpy)."""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, np.uint8)
_COMP[list(b"ACGTN")] = list(b"TGCAN")


def random_seq(rng, n, gc=0.42):
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    return BASES[rng.choice(4, size=n, p=p)]


def revcomp_arr(a):
    return _COMP[a[::-1]]


def mutate(rng, seq, sub=0.04, ins=0.02, dele=0.04):
    """Apply i.i.d. substitutions / insertions / deletions; returns the new uint8 array."""
    n = len(seq)
    r = rng.random(n)
    keep = r >= dele
    is_sub = (r >= dele) & (r < dele + sub)
    out = seq.copy()
    if is_sub.any():
        idx = np.nonzero(is_sub)[0]
        cur = np.searchsorted(BASES, out[idx])
        out[idx] = BASES[(cur + rng.integers(1, 4, size=len(idx))) % 4]
    is_ins = rng.random(n) < ins
    counts = keep.astype(np.int64) + is_ins.astype(np.int64)
    total = int(counts.sum())
    res = np.empty(total, np.uint8)
    pos = np.cumsum(counts) - counts
    k_idx = np.nonzero(keep)[0]
    res[pos[k_idx]] = out[k_idx]
    i_idx = np.nonzero(is_ins)[0]
    res[pos[i_idx] + keep[i_idx].astype(np.int64)] = BASES[rng.integers(0, 4, size=len(i_idx))]
    return res


def simulate_reads(rng, genome_seqs, n_reads, mean_len, sigma=0.5, min_len=500, max_len=150000, err=(0.04, 0.02, 0.04)):
    """-> (list of uint8 arrays, truth array[n,4] = (seq id, start, end, strand))."""
    lens = np.array([len(g) for g in genome_seqs], dtype=np.int64)
    cum = np.cumsum(lens)
    mu = np.log(mean_len) - sigma * sigma / 2
    reads, truth = [], np.zeros((n_reads, 4), np.int64)
    for i in range(n_reads):
        L = int(np.clip(rng.lognormal(mu, sigma), min_len, max_len))
        g = int(np.searchsorted(cum, rng.integers(0, cum[-1]), side="right"))
        L = min(L, int(lens[g]))
        s = int(rng.integers(0, lens[g] - L + 1))
        frag = genome_seqs[g][s:s + L]
        strand = int(rng.integers(0, 2))
        if strand:
            frag = revcomp_arr(frag)
        reads.append(mutate(rng, frag, *err))
        truth[i] = (g, s, s + L, strand)
    return reads, truth
