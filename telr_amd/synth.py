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
def burst_mask(rng, n, burst):
    """bases inside an error burst: bursts start at a rate of burst[0] per base and last burst[1] .. burst[2] bases"""
    st = np.nonzero(rng.random(n, dtype=np.float32) < burst[0])[0]
    d = np.zeros(n + 1, np.int32)
    if len(st):
        ln = rng.integers(burst[1], burst[2] + 1, size=len(st))
        np.add.at(d, st, 1); np.add.at(d, np.minimum(st + ln, n), -1)
    return np.cumsum(d[:n]) > 0


def mutate_bulk(rng, seq, seg_len, sub, ins, dele, burst=None):
    """mutate() over a concatenation of segments; returns (new array, new per-segment lengths).
    burst = (starts per base, shortest, longest, factor): inside a burst every error rate is `factor` times the read's (the
    hard data sets: a stretch of a read where the basecaller lost the signal)."""
    n = len(seq)
    if burst is not None:
        f = np.where(burst_mask(rng, n, burst), np.float32(burst[3]), np.float32(1.0))
        sub, ins, dele = sub * f, ins * f, dele * f
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


def sample_reads_bulk(rng, hap, n_reads, mean_len, sigma, min_len, max_len, err, chunk=512, burst=None):
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
        res, nl = mutate_bulk(rng, frag, L, *err, burst=burst)
        bufs.append(res); out_len[c0:c1] = nl
    buf = np.concatenate(bufs) if bufs else np.zeros(0, np.uint8)
    off = np.cumsum(out_len) - out_len
    truth = np.stack([starts, starts + lens, strand], axis=1)
    return buf, off.astype(np.int64), out_len.astype(np.int32), truth


def make_te_library(rng, n_fam, lo=300, hi=8000):
    lens = np.exp(rng.uniform(np.log(lo), np.log(hi), size=n_fam)).astype(np.int64)
    return [random_seq(rng, int(L), gc=0.45) for L in lens]


# ---------------------------------------------------------------------------------------
# The HARD genome (round 6).  The i.i.d.-plus-dispersed-TE-copies genome above has none of the sequence classes a long-read
# aligner's heuristics exist for (minimap2's occurrence filter, seed rescue, max_chain_skip; NGMLR's candidate voting) and in
# which dm6's own TEs sit: tandem arrays, microsatellites, low-complexity stretches, satellite blocks, segmental duplications.
# `hard` adds them, seeded, and the reads get error bursts.  Defaults: ~4 % of the sequence in tandem arrays (unit 2-500 bp,
# array 0.1-50 kb, copies 0-5 % diverged), ~150 microsatellites per Mb (unit 1-6 bp, 20-300 bp), ~1 % AT-rich / homopolymer-rich
# low-complexity stretches (100-3,000 bp), one segmental duplication per ~6 Mb (10-100 kb, 1-5 % diverged, either orientation),
# a satellite block (unit 5-400 bp, 1-10 kb) within 50-400 bp of 15 % of the spiked insertions, and in the reads one error burst
# per ~20 kb (50-300 bases at 3.5 times the read's error rates).
HARD = dict(tandem_frac=0.04, micro_per_mb=150, lowcx_frac=0.01, segdup_every=6_000_000, sat_ins_frac=0.15, burst=(1.0 / 20000, 50, 300, 3.5))


def _tandem_array(r, unit_len, arr_len, gc, div):
    unit = random_seq_fast(r, unit_len, gc)
    arr = np.tile(unit, arr_len // unit_len + 2)[:arr_len]
    if div > 0:
        arr = mutate(r, arr, div, div / 4, div / 4)
    return arr


def _put(ref, p, seq, lo=0):
    """overwrite ref[p : p + len(seq)] (clipped to the sequence, never into the leading N block) -> (start, end) written"""
    p = max(lo, p); e = min(len(ref), p + len(seq))
    if e <= p:
        return p, p
    ref[p:e] = seq[:e - p]
    return p, e


def harden_sequence(r, ref, lo=0, gc=0.42, hard=None):
    """tandem arrays, microsatellites, low-complexity stretches and segmental duplications written over `ref` (in place) from
    generator `r`; -> [(start, end, kind)]"""
    hard = HARD if hard is None or hard is True else hard
    L = len(ref)
    feats = []
    if L - lo < 50000:
        return feats
    covered = 0
    while covered < hard["tandem_frac"] * (L - lo):
        unit_len = int(np.exp(r.uniform(np.log(2), np.log(500))))
        arr_len = int(np.exp(r.uniform(np.log(100), np.log(50000))))
        arr = _tandem_array(r, unit_len, max(arr_len, 2 * unit_len), gc, float(r.uniform(0, 0.05)))
        s, e = _put(ref, int(r.integers(lo, L - len(arr))), arr, lo)
        feats.append((s, e, "tandem")); covered += e - s
    for _ in range(int(hard["micro_per_mb"] * (L - lo) / 1e6)):
        arr = _tandem_array(r, int(r.integers(1, 7)), int(r.integers(20, 301)), 0.5, float(r.uniform(0, 0.03)))
        s, e = _put(ref, int(r.integers(lo, L - len(arr))), arr, lo)
        feats.append((s, e, "micro"))
    covered = 0
    while covered < hard["lowcx_frac"] * (L - lo):
        n = int(np.exp(r.uniform(np.log(100), np.log(3000))))
        if r.random() < 0.5:
            seq = random_seq_fast(r, n, float(r.uniform(0.03, 0.2)))                     # AT-rich
        else:                                                                           # runs of homopolymers, 3-25 bases each
            runs = r.integers(3, 26, size=n // 3 + 1)
            seq = np.repeat(BASES[r.integers(0, 4, size=len(runs))], runs)[:n]
        s, e = _put(ref, int(r.integers(lo, L - n)), seq, lo)
        feats.append((s, e, "lowcx")); covered += e - s
    for _ in range(max(1 if L - lo >= 1_000_000 else 0, int((L - lo) // hard["segdup_every"]))):
        n = int(np.exp(r.uniform(np.log(10000), np.log(100000))))
        src = int(r.integers(lo, L - n)); dst = int(r.integers(lo, L - n - n // 20))
        if abs(src - dst) < 2 * n:
            continue
        d = float(r.uniform(0.01, 0.05))
        cp = mutate(r, ref[src:src + n], d * 0.8, d * 0.1, d * 0.1)
        if r.integers(0, 2):
            cp = revcomp_arr(cp)
        s, e = _put(ref, dst, cp, lo)
        feats.append((src, src + n, "segdup_src")); feats.append((s, e, "segdup"))
    return feats


def satellites_at_sites(r, ref, sites, tsd_max=8, gc=0.42, hard=None, keep_clear=300):
    """a satellite block next to a fraction of the insertion sites (`sites`: sorted positions on this sequence): unit 5-400 bp,
    1-10 kb, 0-3 % diverged copies, starting 50-400 bp after the site's duplication or ending 50-400 bp before the site; never
    over a site itself nor within `keep_clear` bases of a neighbouring one.  -> [(start, end, 'satellite', site)]"""
    hard = HARD if hard is None or hard is True else hard
    feats = []
    sites = [int(x) for x in sites]
    for k, p in enumerate(sites):
        if r.random() >= hard["sat_ins_frac"]:
            continue
        gap = int(r.integers(50, 401))
        arr = _tandem_array(r, int(np.exp(r.uniform(np.log(5), np.log(400)))), int(np.exp(r.uniform(np.log(1000), np.log(10000)))), gc, float(r.uniform(0, 0.03)))
        if r.integers(0, 2):          # downstream of the site
            s = p + tsd_max + gap
            room = (sites[k + 1] - keep_clear if k + 1 < len(sites) else len(ref)) - s
        else:                          # upstream
            room = p - gap - (sites[k - 1] + tsd_max + keep_clear if k > 0 else 0)
            s = p - gap - min(len(arr), max(0, room))
        n = min(len(arr), room)
        if n < 200 or s < 0:
            continue
        ref[s:s + n] = arr[:n]
        feats.append((s, s + n, "satellite", p))
    return feats


def make_stage1_dataset(seed=20261002, genome_len=23513712, n_reads=10000, total_bases=470_000_000, n_ins=200,
                        n_fam=127, te_frac=0.15, gc=0.42, err=(0.04, 0.02, 0.04), sigma=0.6, read_seed=None, hard=None):
    """BASELINE.json configs[1]: synthetic chr2L-sized genome + ONT-like reads + spiked TE insertions.

    Returns dict(ref=uint8 array, library=[arrays], reads=(buf, off, len), insertions=[(pos, fam, strand, tsd, af)],
    truth=(hap id, start, end, strand) per read in haplotype coordinates)."""
    rng = np.random.default_rng(seed)
    ref = random_seq(rng, genome_len, gc)
    lib = make_te_library(rng, n_fam)
    # reference TE copies: diverged, 5'-truncated copies over ~te_frac of the genome
    covered = 0
    te_copies = []                  # (start, end) of the reference's TE-derived stretches (reported, not used by the generator)
    while covered < te_frac * genome_len:
        f = lib[int(rng.integers(0, n_fam))]
        cut = int(rng.integers(0, max(1, len(f) // 2)))
        cp = mutate(rng, f[cut:], float(rng.uniform(0, 0.15)), 0.0, 0.0)
        if rng.integers(0, 2):
            cp = revcomp_arr(cp)
        p = int(rng.integers(0, genome_len - len(cp)))
        ref[p:p + len(cp)] = cp
        covered += len(cp)
        te_copies.append((p, p + len(cp)))
    hard_feats = []
    if hard:
        hard_feats = harden_sequence(np.random.default_rng([seed, 7]), ref, 0, gc, hard)
    # spiked non-reference insertions, >= 5 kb apart
    sites = np.sort(rng.choice(np.arange(5000, genome_len - 5000, 5000), size=n_ins, replace=False)) + rng.integers(0, 2000, size=n_ins)
    if hard:
        hard_feats += satellites_at_sites(np.random.default_rng([seed, 8]), ref, sites, gc=gc, hard=hard)
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
    burst = (HARD if hard is True else hard)["burst"] if hard else None
    rA = sample_reads_bulk(rng, haps[0], nA, mean_len, sigma, 500, 150000, err, burst=burst)
    rB = sample_reads_bulk(rng, haps[1], n_reads - nA, mean_len, sigma, 500, 150000, err, burst=burst)
    buf = np.concatenate([rA[0], rB[0]])
    ln = np.concatenate([rA[2], rB[2]])
    off = np.cumsum(ln.astype(np.int64)) - ln
    truth = np.concatenate([np.c_[np.zeros(nA, np.int64), rA[3]], np.c_[np.ones(n_reads - nA, np.int64), rB[3]]])
    return dict(ref=ref, library=lib, reads=(buf, off.astype(np.int64), ln.astype(np.int32)), insertions=ins, truth=truth, haps=haps, te_copies=te_copies, hard_features=hard_feats)


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


# ---------------------------------------------------------------------------------------
# Multi-chromosome data sets (BASELINE configs[2]-[4]; SURVEY.md 8d recipe).  Everything is derived from
# (seed, object index) so that any rank can materialise any block of reads on its own: the read PLAN (chromosome,
# haplotype, start, length, strand of every read) is cheap and made everywhere, the bases of a block of reads come from
# a generator seeded with (seed, block index) whatever the world size.
DM6_ARMS = [("chr2L", 23513712), ("chr2R", 25286936), ("chr3L", 28110227), ("chr3R", 32079331),
            ("chr4", 1348131), ("chrX", 23542271), ("chrY", 3667352), ("chrM", 19524)]          # sum 137,567,484
CHR22 = [("chr22", 50818468)]
N_HAP = 4            # allele frequencies 0.25 / 0.5 / 1.0 = insertion present on 1 / 2 / 4 of four haplotypes
READ_BLOCK = 256     # reads per generation block (the unit dealt to ranks in the strong-scaling bench)


def random_seq_fast(rng, n, gc=0.42):
    u = rng.random(n, dtype=np.float32)
    a = (1 - gc) / 2
    code = (u >= a).astype(np.uint8) + (u >= a + gc / 2) + (u >= a + gc)
    return BASES[code]


def _pool_map(fn, items, threads):
    if threads <= 1 or len(items) <= 1:
        return [fn(x) for x in items]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(fn, items))


def make_genome(seed, chroms, n_fam=127, te_frac=0.15, gc=0.42, n_ins=1000, lead_n=0, afs=(0.25, 0.5, 1.0), threads=1, hard=None):
    """-> dict(names, ref=[uint8 arrays], library, insertions=[(chrom id, pos, fam, strand, tsd, af)] sorted,
    haps[h][c] = uint8 arrays, hap_ins[h][c] = (ref positions, cumulative shift after each insertion present on h)).
    lead_n: length of a leading N block of every chromosome (chr22's 11 Mb)."""
    rng = np.random.default_rng([seed, 0])
    lib = make_te_library(rng, n_fam)
    names = [c[0] for c in chroms]

    def one_chrom(ci):
        r = np.random.default_rng([seed, 1, ci])
        L = chroms[ci][1]
        ref = random_seq_fast(r, L, gc)
        lo = min(lead_n, max(0, L - 100000))
        covered = 0
        while covered < te_frac * (L - lo) and L - lo > 20000:
            f = lib[int(r.integers(0, n_fam))]
            cut = int(r.integers(0, max(1, len(f) // 2)))
            cp = mutate(r, f[cut:], float(r.uniform(0, 0.15)), 0.0, 0.0)
            if r.integers(0, 2):
                cp = revcomp_arr(cp)
            if len(cp) >= L - lo:
                continue
            p = int(r.integers(lo, L - len(cp)))
            ref[p:p + len(cp)] = cp
            covered += len(cp)
        feats = harden_sequence(np.random.default_rng([seed, 7, ci]), ref, lo, gc, hard) if hard else []
        if lo:
            ref[:lo] = ord("N")
        return ref, feats

    made = _pool_map(one_chrom, list(range(len(chroms))), threads)
    refs = [m[0] for m in made]
    hard_feats = [[(ci,) + f for f in m[1]] for ci, m in enumerate(made)]
    # insertion sites: slots of a 5-kb grid over all chromosomes (>= 3 kb apart after the jitter), outside the N block
    slots_c, slots_p = [], []
    for ci, (_, L) in enumerate(chroms):
        g = np.arange(max(5000, lead_n + 5000), L - 7000, 5000)
        slots_c.append(np.full(len(g), ci)); slots_p.append(g)
    slots_c = np.concatenate(slots_c); slots_p = np.concatenate(slots_p)
    n_ins = min(n_ins, len(slots_p))
    pick = np.sort(rng.choice(len(slots_p), size=n_ins, replace=False))
    ins = []
    for k in pick:
        ins.append((int(slots_c[k]), int(slots_p[k] + rng.integers(0, 2000)), int(rng.integers(0, n_fam)), int(rng.integers(0, 2)),
                    int(rng.integers(4, 9)), float(rng.choice(afs))))
    ins.sort()
    if hard:          # satellite blocks next to a share of the insertion sites (before the haplotypes are cut from the reference)
        for ci in range(len(chroms)):
            sites = [p for (c, p, *_rest) in ins if c == ci]
            hard_feats[ci] += [(ci,) + f for f in satellites_at_sites(np.random.default_rng([seed, 8, ci]), refs[ci], sites, gc=gc, hard=hard)]

    def build(hc):
        h, ci = hc
        ref = refs[ci]
        parts, last, pos, shift, tot = [], 0, [], [], 0
        for (c, p, fam, strand, tsd, af) in ins:
            if c != ci or h >= round(af * N_HAP):
                continue
            te = lib[fam] if not strand else revcomp_arr(lib[fam])
            parts += [ref[last:p + tsd], te, ref[p:p + tsd]]
            last = p + tsd
            tot += len(te) + tsd
            pos.append(p); shift.append(tot)
        parts.append(ref[last:])
        return (np.concatenate(parts) if len(parts) > 1 else ref), (np.array(pos, np.int64), np.array(shift, np.int64))

    built = _pool_map(build, [(h, ci) for h in range(N_HAP) for ci in range(len(chroms))], threads)
    haps = [[built[h * len(chroms) + ci][0] for ci in range(len(chroms))] for h in range(N_HAP)]
    hap_ins = [[built[h * len(chroms) + ci][1] for ci in range(len(chroms))] for h in range(N_HAP)]
    return dict(names=names, ref=refs, library=lib, insertions=ins, haps=haps, hap_ins=hap_ins, seed=seed, hard_features=[f for fs in hard_feats for f in fs])


def hap_to_ref(g, h, ci, x):
    """reference coordinate of haplotype coordinate(s) x (positions inside an inserted element map to its site)"""
    pos, shift = g["hap_ins"][h][ci]
    x = np.asarray(x, np.int64)
    if len(pos) == 0:
        return x
    # hap coordinate at which insertion k starts = pos[k] + tsd + shift[k-1]; solve by searching the shifted starts
    prev = np.r_[0, shift[:-1]]
    hstart = pos + prev                     # (tsd is part of the shift of the same insertion; an error of < 10 bp is fine for truth checks)
    k = np.searchsorted(hstart, x, side="right") - 1
    sh = np.where(k >= 0, shift[np.maximum(k, 0)], 0)
    inside = (k >= 0) & (x < hstart[np.maximum(k, 0)] + (shift[np.maximum(k, 0)] - prev[np.maximum(k, 0)]))
    return np.where(inside, pos[np.maximum(k, 0)], x - sh)


def plan_reads(g, coverage=30.0, mean_len=9000, sigma=0.6, min_len=500, max_len=150000, read_seed=None):
    """The read plan: arrays (chrom, hap, start, length, strand) of every read, sampled until `coverage` x the
    reference bases.  Reads start uniformly on a chromosome chosen by length and never cross its end."""
    rng = np.random.default_rng([g["seed"] if read_seed is None else read_seed, 2])
    clen = np.array([len(r) for r in g["ref"]], np.int64)
    target = int(coverage * clen.sum())
    mu = np.log(mean_len) - sigma * sigma / 2
    n0 = int(target / mean_len * 1.3) + 64
    lens = np.clip(rng.lognormal(mu, sigma, size=n0), min_len, max_len).astype(np.int64)
    chrom = np.searchsorted(np.cumsum(clen), rng.integers(0, clen.sum(), size=n0), side="right")
    hap = rng.integers(0, N_HAP, size=n0)
    hlen = np.array([[len(g["haps"][h][c]) for c in range(len(clen))] for h in range(N_HAP)], np.int64)
    L = np.minimum(lens, hlen[hap, chrom])
    n = int(np.searchsorted(np.cumsum(L), target, side="left")) + 1
    n = min(n, n0)
    L, chrom, hap = L[:n], chrom[:n], hap[:n]
    start = (rng.random(n) * (hlen[hap, chrom] - L + 1)).astype(np.int64)
    strand = rng.integers(0, 2, size=n).astype(np.int64)
    return dict(chrom=chrom.astype(np.int32), hap=hap.astype(np.int32), start=start, length=L, strand=strand.astype(np.int8),
                n=n, n_blocks=(n + READ_BLOCK - 1) // READ_BLOCK, seed=g["seed"] if read_seed is None else read_seed)


def block_bases(plan):
    """planned (error-free) bases per block, for dealing blocks to ranks"""
    nb = plan["n_blocks"]
    pad = np.zeros(nb * READ_BLOCK, np.int64); pad[:plan["n"]] = plan["length"]
    return pad.reshape(nb, READ_BLOCK).sum(axis=1)


_MAT = {}        # state shared with forked workers (set before the pool starts; read-only afterwards)


def _mat_block(b):
    g_haps, plan, err = _MAT["haps"], _MAT["plan"], _MAT["err"]
    r0, r1 = b * READ_BLOCK, min(plan["n"], (b + 1) * READ_BLOCK)
    rng = np.random.default_rng([plan["seed"], 3, int(b)])
    L = plan["length"][r0:r1]
    frags = []
    for i in range(r0, r1):
        s = int(plan["start"][i]); f = g_haps[plan["hap"][i]][plan["chrom"][i]][s:s + int(plan["length"][i])]
        frags.append(_COMP[f[::-1]] if plan["strand"][i] else f)
    frag = np.concatenate(frags) if frags else np.zeros(0, np.uint8)
    return mutate_bulk(rng, frag, L, *err, burst=_MAT.get("burst"))


def materialize_reads(g, plan, blocks=None, err=(0.04, 0.02, 0.04), procs=1, burst=None):
    """Bases of the reads of `blocks` (default all), block by block with generator (seed, 3, block) -- the same
    bases whatever the world size or the worker count.  procs > 1: forked worker processes (call before the
    process touches the GPU).  -> (buf, off, len, read_ids): read_ids[i] = index of sequence i in the plan."""
    blocks = np.arange(plan["n_blocks"]) if blocks is None else np.asarray(blocks)
    _MAT.update(haps=g["haps"], plan=plan, err=err, burst=burst)
    items = [int(b) for b in blocks]
    if procs > 1 and len(items) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(procs) as pool:
            parts = pool.map(_mat_block, items, chunksize=max(1, min(16, len(items) // (procs * 4))))
    else:
        parts = [_mat_block(b) for b in items]
    _MAT.clear()
    ln = np.concatenate([p[1] for p in parts]).astype(np.int32) if parts else np.zeros(0, np.int32)
    buf = np.concatenate([p[0] for p in parts]) if parts else np.zeros(0, np.uint8)
    del parts
    off = np.cumsum(ln.astype(np.int64)) - ln
    ids = np.concatenate([np.arange(b * READ_BLOCK, min(plan["n"], (b + 1) * READ_BLOCK)) for b in blocks]) if len(blocks) else np.zeros(0, np.int64)
    return buf, off.astype(np.int64), ln, ids


def make_loci(g, n_loci=None, seed=7, flank=(8000, 15000)):
    """Per-locus inputs of the stage 3/4 bundle when Sniffles / wtdbg2 are unavailable: contig = the
    insertion haplotype +-(8-15) kb around the site with 0.5 % residual error, ALT sequence = the inserted element
    with 5 % error.  The window reads are NOT given here: they come from the stage-1 records
    (telr_assembly.window_reads), as in the reference."""
    rng = np.random.default_rng([g["seed"], 4, seed])
    lib, ins = g["library"], g["insertions"]
    order = np.arange(len(ins)) if n_loci is None or n_loci >= len(ins) else np.sort(rng.permutation(len(ins))[:n_loci])
    loci = []
    for k in order:
        ci, p, fam, strand, tsd, af = ins[k]
        hap0 = g["haps"][0][ci]
        pos, shift = g["hap_ins"][0][ci]
        j = int(np.searchsorted(pos, p))
        a = p + tsd + (int(shift[j - 1]) if j > 0 else 0)         # first base of the element on haplotype 0
        te_len = len(lib[fam])
        lo, hi = int(rng.integers(*flank)), int(rng.integers(*flank))
        s, e = max(0, a - lo), min(len(hap0), a + te_len + hi)
        contig = mutate(rng, hap0[s:e], 0.003, 0.001, 0.001)
        te = lib[fam] if not strand else revcomp_arr(lib[fam])
        alt = mutate(rng, te, 0.03, 0.01, 0.01)
        name = g["names"][ci]
        loci.append({"name": "%s_%d_%d" % (name, p, p + 1), "chrom": name, "start": p, "end": p + 1, "contig": bytes(contig).decode(),
                     "alt": bytes(alt).decode(), "truth": {"chrom": name, "pos": p, "family": "fam%d" % fam, "strand": "+-"[strand], "tsd": tsd, "af": af}})
    return loci
