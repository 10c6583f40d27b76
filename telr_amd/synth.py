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


# ---------------------------------------------------------------------------------------
# bulk generators used by bench.py (vectorised over whole read sets)
def mutate_bulk(rng, seq, seg_len, sub, ins, dele):
    """mutate() over a concatenation of segments; returns (new array, new per-segment lengths)."""
    n = len(seq)
    r = rng.random(n, dtype=np.float32)
    keep = r >= dele
    is_sub = keep & (r < dele + sub)
    out = seq.copy()
    idx = np.nonzero(is_sub)[0]
    if len(idx):
        lut = np.zeros(256, np.uint8); lut[BASES] = np.arange(4, dtype=np.uint8)
        out[idx] = BASES[(lut[out[idx]] + rng.integers(1, 4, size=len(idx), dtype=np.uint8)) % 4]
    is_ins = rng.random(n, dtype=np.float32) < ins
    counts = keep.astype(np.int8) + is_ins.astype(np.int8)
    csum = np.cumsum(counts, dtype=np.int64)
    total = int(csum[-1]) if n else 0
    pos = csum - counts
    res = np.empty(total, np.uint8)
    k_idx = np.nonzero(keep)[0]
    res[pos[k_idx]] = out[k_idx]
    i_idx = np.nonzero(is_ins)[0]
    res[pos[i_idx] + keep[i_idx]] = BASES[rng.integers(0, 4, size=len(i_idx))]
    seg_end = np.cumsum(seg_len, dtype=np.int64)
    seg_start = seg_end - seg_len
    new_end = np.where(seg_len > 0, csum[np.maximum(seg_end - 1, 0)], 0)
    new_start = np.where(seg_start > 0, csum[np.maximum(seg_start - 1, 0)], 0)
    new_len = np.where(seg_len > 0, new_end - new_start, 0)
    return res, new_len.astype(np.int64)


def sample_reads_bulk(rng, hap, n_reads, mean_len, sigma, min_len, max_len, err, chunk=512):
    """Reads from one haplotype (uint8 array). -> (buf, off, len, truth[n,3]=(start,end,strand))"""
    G = len(hap)
    mu = np.log(mean_len) - sigma * sigma / 2
    lens = np.clip(rng.lognormal(mu, sigma, size=n_reads), min_len, min(max_len, G)).astype(np.int64)
    starts = (rng.random(n_reads) * (G - lens + 1)).astype(np.int64)
    strand = rng.integers(0, 2, size=n_reads).astype(np.int64)
    bufs, out_len = [], np.zeros(n_reads, np.int64)
    for c0 in range(0, n_reads, chunk):
        c1 = min(n_reads, c0 + chunk)
        L = lens[c0:c1]
        tot = int(L.sum())
        seg_off = np.cumsum(L) - L
        rid = np.repeat(np.arange(c1 - c0), L)
        within = np.arange(tot, dtype=np.int64) - seg_off[rid]
        st = strand[c0:c1][rid]
        idx = np.where(st == 1, starts[c0:c1][rid] + L[rid] - 1 - within, starts[c0:c1][rid] + within)
        frag = hap[idx]
        frag = np.where(st == 1, _COMP[frag], frag)
        res, nl = mutate_bulk(rng, frag, L, *err)
        bufs.append(res); out_len[c0:c1] = nl
    buf = np.concatenate(bufs) if bufs else np.zeros(0, np.uint8)
    off = np.cumsum(out_len) - out_len
    truth = np.stack([starts, starts + lens, strand], axis=1)
    return buf, off.astype(np.int64), out_len.astype(np.int32), truth


def make_te_library(rng, n_fam, lo=300, hi=8000):
    lens = np.exp(rng.uniform(np.log(lo), np.log(hi), size=n_fam)).astype(np.int64)
    return [random_seq(rng, int(L), gc=0.45) for L in lens]


def make_stage1_dataset(seed=20261002, genome_len=23513712, n_reads=10000, total_bases=470_000_000, n_ins=200,
                        n_fam=127, te_frac=0.15, gc=0.42, err=(0.04, 0.02, 0.04), sigma=0.6, read_seed=None):
    """BASELINE.json configs[1]: synthetic chr2L-sized genome + ONT-like reads + spiked TE insertions.

    Returns dict(ref=uint8 array, library=[arrays], reads=(buf, off, len), insertions=[(pos, fam, strand, tsd, af)],
    truth=(hap id, start, end, strand) per read in haplotype coordinates)."""
    rng = np.random.default_rng(seed)
    ref = random_seq(rng, genome_len, gc)
    lib = make_te_library(rng, n_fam)
    # reference TE copies: diverged, 5'-truncated copies over ~te_frac of the genome
    covered = 0
    while covered < te_frac * genome_len:
        f = lib[int(rng.integers(0, n_fam))]
        cut = int(rng.integers(0, max(1, len(f) // 2)))
        cp = mutate(rng, f[cut:], float(rng.uniform(0, 0.15)), 0.0, 0.0)
        if rng.integers(0, 2):
            cp = revcomp_arr(cp)
        p = int(rng.integers(0, genome_len - len(cp)))
        ref[p:p + len(cp)] = cp
        covered += len(cp)
    # spiked non-reference insertions, >= 5 kb apart
    sites = np.sort(rng.choice(np.arange(5000, genome_len - 5000, 5000), size=n_ins, replace=False)) + rng.integers(0, 2000, size=n_ins)
    ins = []
    for p in sites:
        ins.append((int(p), int(rng.integers(0, n_fam)), int(rng.integers(0, 2)), int(rng.integers(4, 9)), float(rng.choice([0.5, 1.0]))))

    def build_hap(which):
        parts, last = [], 0
        for (p, fam, strand, tsd, af) in ins:
            if af < 1.0 and which == 1:
                continue
            te = lib[fam] if not strand else revcomp_arr(lib[fam])
            parts += [ref[last:p + tsd], te, ref[p:p + tsd]]     # target-site duplication
            last = p + tsd
        parts.append(ref[last:])
        return np.concatenate(parts)

    haps = [build_hap(0), build_hap(1)]
    if read_seed is not None:
        rng = np.random.default_rng(read_seed)
    mean_len = total_bases / n_reads
    nA = n_reads // 2
    rA = sample_reads_bulk(rng, haps[0], nA, mean_len, sigma, 500, 150000, err)
    rB = sample_reads_bulk(rng, haps[1], n_reads - nA, mean_len, sigma, 500, 150000, err)
    buf = np.concatenate([rA[0], rB[0]])
    ln = np.concatenate([rA[2], rB[2]])
    off = np.cumsum(ln.astype(np.int64)) - ln
    truth = np.concatenate([np.c_[np.zeros(nA, np.int64), rA[3]], np.c_[np.ones(n_reads - nA, np.int64), rB[3]]])
    return dict(ref=ref, library=lib, reads=(buf, off.astype(np.int64), ln.astype(np.int32)), insertions=ins, truth=truth, haps=haps)


def make_loci_from_dataset(d, n_loci, seed=7, flank=(8000, 15000), reads_cap=60, window=1000):
    """Per-locus inputs for the stage 3/4 bundle when Sniffles / wtdbg2 are unavailable:
    contig = true insertion haplotype +-(8-15) kb around the site with 0.5 % residual error, ALT sequence =
    true insertion with 5 % error, window reads = the simulated reads overlapping +-1 kb of the site (truth)."""
    rng = np.random.default_rng(seed)
    ref, lib, ins = d["ref"], d["library"], d["insertions"]
    buf, off, ln = d["reads"]
    truth = d["truth"]
    hap0 = d["haps"][0]
    # insertion coordinates on haplotype 0 (every insertion is present there)
    shift = 0
    loci = []
    order = rng.permutation(len(ins))[:n_loci]
    pos_h0 = []
    for (p, fam, strand, tsd, af) in ins:
        pos_h0.append(p + shift + tsd)            # first TE base on hap0
        shift += len(lib[fam]) + tsd
    # hap1 coordinates: only AF==1 insertions are present
    shift1, pos_h1 = 0, []
    for (p, fam, strand, tsd, af) in ins:
        pos_h1.append(p + shift1)
        if af >= 1.0:
            shift1 += len(lib[fam]) + tsd
    for k in sorted(order):
        p, fam, strand, tsd, af = ins[k]
        te_len = len(lib[fam])
        a = pos_h0[k]
        lo, hi = int(rng.integers(*flank)), int(rng.integers(*flank))
        s, e = max(0, a - lo), min(len(hap0), a + te_len + hi)
        contig = mutate(rng, hap0[s:e], 0.003, 0.001, 0.001)
        te = lib[fam] if not strand else revcomp_arr(lib[fam])
        alt = mutate(rng, te, 0.03, 0.01, 0.01)
        # reads overlapping the window on their own haplotype
        sel = []
        for hap_id, centre in ((0, a), (1, pos_h1[k])):
            m = (truth[:, 0] == hap_id) & (truth[:, 1] < centre + window) & (truth[:, 2] > centre - window)
            sel.extend(np.nonzero(m)[0].tolist())
        sel = sel[:reads_cap]
        reads = [bytes(buf[off[i]:off[i] + ln[i]]).decode() for i in sel]
        loci.append({"name": "chr2L_%d_%d" % (p, p + 1), "contig": bytes(contig).decode(), "alt": bytes(alt).decode(), "reads": reads, "read_idx": list(sel),
                     "truth": {"pos": p, "family": "fam%d" % fam, "strand": "+-"[strand], "tsd": tsd, "af": af}})
    return loci
