"""Randomised HIP-vs-oracle parity: random genomes, read sets and option mixes, every stage compared bit for bit
(tests/test_gpu_parity.py: compare_all).  usage: [FUZZ_BIG=1] [FUZZ_HARD=1] python tests/fuzz_parity.py [iterations] [seed]   (FUZZ_BIG: Mb-size genomes, reads of 8-40 kb; FUZZ_HARD: repeat-rich targets, reads with error bursts)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # test infrastructure: the oracle is the checker
os.environ["TELR_DEBUG"] = "1"
import numpy as np
from telr_amd import synth
from telr_amd.aligner import Engine
from telr_amd.presets import preset



def harden(rng, genome, reads):
    """round 6: the HARD sequence classes of telr_amd/synth.py in fuzz size -- tandem arrays, microsatellites, low-complexity stretches,
    (on Mb-size targets) a segmental duplication -- written over the targets, and error bursts in a third of the reads"""
    for g in genome:
        if len(g) >= 300000:
            synth.harden_sequence(rng, g, 0, 0.42, dict(synth.HARD, tandem_frac=0.08, micro_per_mb=600, lowcx_frac=0.03, segdup_every=400_000))
        else:
            for _ in range(int(rng.integers(1, 5))):
                arr = synth._tandem_array(rng, int(np.exp(rng.uniform(np.log(2), np.log(200)))), int(rng.integers(100, 3000)), 0.42, float(rng.uniform(0, 0.05)))
                synth._put(g, int(rng.integers(0, max(1, len(g) - len(arr)))), arr)
            for _ in range(int(rng.integers(2, 12))):
                arr = synth._tandem_array(rng, int(rng.integers(1, 7)), int(rng.integers(20, 300)), 0.5, 0.0)
                synth._put(g, int(rng.integers(0, max(1, len(g) - len(arr)))), arr)
            n = int(rng.integers(100, 800))
            synth._put(g, int(rng.integers(0, max(1, len(g) - n))), synth.random_seq_fast(rng, n, 0.08))
    return genome


def burst(rng, reads):
    for ri in range(len(reads)):
        r = reads[ri]
        if len(r) > 600 and rng.random() < 0.33:
            L = int(rng.integers(50, 300)); p = int(rng.integers(100, len(r) - L - 100))
            reads[ri] = np.concatenate([r[:p], synth.mutate(rng, r[p:p + L], 0.12, 0.06, 0.12), r[p + L:]])
    return reads


def draw_case(seed, big=False, sv=False, presets=None, hard=False):
    """one random configuration -> (preset name, io, mo, genome, reads, qtarget, edge_repeats).
    hard: tandem arrays / microsatellites / low-complexity stretches / segmental duplications in the targets, error bursts in the reads
    (FUZZ_HARD=1): reads that sit in repeats -- many equal-score chains, anchors past the LDS sort's limit, extensions through arrays.
    sv: half of the reads carry one to three large insertions / deletions (30-600 bases): wide bands, long gap runs -- the
    convex cost's re-biased int16 classes, their multi-wave form and the int32 fall-back (FUZZ_SV=1; FUZZ_PRESETS=a,b restricts
    the presets drawn)"""
    rng = np.random.default_rng(seed)
    names = presets or ["map-ont", "map-ont", "map-pb", "asm10", "ngmlr-ont", "ngmlr-pacbio"]
    pname = names[int(rng.integers(0, len(names)))]
    io, mo = preset(pname)
    ntg = int(rng.integers(1, 4))
    genome = [synth.random_seq(rng, int(rng.integers(300000, 1500000) if big else rng.integers(20000, 120000))) for _ in range(ntg)]
    te = synth.random_seq(rng, int(rng.integers(500, 4000)))
    for g in genome:                                   # repeats: occurrence filter, secondary chains
        for _ in range(int(rng.integers(0, 8))):
            p = int(rng.integers(0, len(g) - len(te)))
            g[p:p + len(te)] = synth.mutate(rng, te, float(rng.uniform(0, 0.08)), 0.0, 0.0)[:len(te)]
    if hard:
        harden(rng, genome, None)
    if rng.random() < 0.3:
        g = genome[0]; p = int(rng.integers(0, len(g) - 300)); g[p:p + int(rng.integers(1, 300))] = ord("N")
    # round 4: targets that BEGIN (or end) inside a repeat copy, with homopolymer runs at the very start -- the class of the
    # chain-start bug of round 3 (HPC minimizers + per-query target + a chain at base 0 of the target)
    edge_repeats = rng.random() < 0.35
    if edge_repeats:
        for g in genome:
            if rng.random() < 0.7:
                cut = int(rng.integers(0, len(te) // 2))
                g[:len(te) - cut] = synth.mutate(rng, te, float(rng.uniform(0, 0.05)), 0.0, 0.0)[cut:len(te)]
            if rng.random() < 0.5:
                g[:int(rng.integers(2, 12))] = g[0]                      # a homopolymer run at base 0
            if rng.random() < 0.5:
                cut = int(rng.integers(1, len(te) // 2))
                g[len(g) - cut:] = synth.mutate(rng, te, float(rng.uniform(0, 0.05)), 0.0, 0.0)[:cut]
    err = float(rng.uniform(0.0, 0.07))
    reads, truth = synth.simulate_reads(rng, genome, int(rng.integers(5, 70)), int(rng.integers(8000, 40000) if big else rng.integers(400, 9000)), err=(err, err / 2, err))
    truth = [int(t[0]) for t in truth]
    if hard:
        reads = burst(rng, reads)
    if sv:
        for ri in range(len(reads)):
            if rng.random() < 0.5 and len(reads[ri]) > 1500:
                r = reads[ri]
                for _ in range(int(rng.integers(1, 4))):
                    L = int(rng.integers(30, 600)); p = int(rng.integers(300, max(301, len(r) - 300 - L)))
                    r = np.concatenate([r[:p], r[p + L:]]) if rng.random() < 0.5 else np.concatenate([r[:p], synth.random_seq(rng, L), r[p:]])
                reads[ri] = r
    if edge_repeats:                                  # reads that start exactly at base 0 / end at the last base of a target
        for _ in range(int(rng.integers(1, 6))):
            gi = int(rng.integers(0, ntg)); g = genome[gi]
            L = int(min(len(g), rng.integers(300, 6000)))
            frag = g[:L] if rng.random() < 0.6 else g[len(g) - L:]
            if rng.random() < 0.5:
                frag = synth.revcomp_arr(frag)
            reads.append(synth.mutate(rng, frag, err, err / 2, err)); truth.append(gi)
    for _ in range(int(rng.integers(0, 3))):           # odd ones: tiny, with Ns, empty
        reads.append(synth.random_seq(rng, int(rng.integers(0, 40)))); truth.append(int(rng.integers(0, ntg)))
    if reads and rng.random() < 0.5:
        r = reads[int(rng.integers(0, len(reads)))]
        if len(r) > 200:
            r[50:50 + int(rng.integers(1, 100))] = ord("N")
    # option mix
    if pname != "map-pb" and not pname.startswith("ngmlr") and rng.random() < 0.5:
        io.k = int(rng.integers(11, 22)); io.w = int(rng.choice([5, 10, 10, 12, 19]))
        from telr_amd.presets import _gap_q8
        mo.chain_gap_q8 = _gap_q8(io.k)
    mo.chain_lookback = int(rng.choice([64, 128, 256]))
    mo.fill_band_q4 = int(rng.integers(1, 17)); mo.fill_margin = int(rng.integers(0, 6))
    if rng.random() < 0.25:                             # gap costs on both sides of the one-piece rule of the packed cell ((D-1)(e-e2) < q2-q)
        mo.q2 = int(mo.q + rng.integers(0, 30)); mo.e2 = int(rng.integers(1, mo.e + 1))
    per_target = ntg > 1 and rng.random() < 0.2          # ranked per target, per-target occurrence cut-offs
    if per_target:
        mo.flags |= 2
    # round 4: per-query targets (the S4 / S6 / polishing call shape) with EVERY preset, the HPC one included; mostly the
    # target of origin, sometimes another one, sometimes -1 (unrestricted)
    qtarget = None
    if not per_target and rng.random() < (0.5 if edge_repeats else 0.25):
        qtarget = np.array([t if rng.random() < 0.8 else int(rng.integers(-1, ntg)) for t in truth], dtype=np.int32)
    mo.min_ksw_len = int(rng.choice([50, 100, 200, 400]))
    mo.bw = int(rng.choice([100, 500, 2000])); mo.max_gap = int(rng.choice([1000, 5000, 10000]))
    mo.best_n = int(rng.integers(1, 8)); mo.secondary = int(rng.integers(0, 2))
    mo.chain_skip_q8 = int(rng.choice([0, 0, 0, 3]))
    mo.ext_max = int(rng.choice([256, 2048])); mo.zdrop = int(rng.choice([100, 400]))
    mo.ext_band = int(rng.choice([31, 31, 31, 63, 63, 127, 20, 45, 95]))      # round 5: the extension band is a preset parameter (classes 18 / 23 / 24)
    if mo.bw_long > mo.bw and mo.ext_band > 31:
        mo.ext_band = 31                                 # (the long join's two-band fills are cut to the 64-diagonal extension band: telr_map refuses wider)
    return pname, io, mo, genome, reads, qtarget, edge_repeats


def run(eng, n_iter, seed0, big=False, sv=False, presets=None, hard=False):
    from test_gpu_parity import compare_all
    for it in range(n_iter):
        pname, io, mo, genome, reads, qtarget, edge_repeats = draw_case(seed0 * 1000 + it, big, sv, presets, hard)
        try:
            compare_all(eng, genome, reads, io, mo, qtarget=qtarget)
        except Exception as e:
            print("FAIL iteration", it, "seed", seed0 * 1000 + it, pname, "k", io.k, "w", io.w, "lookback", mo.chain_lookback, "q4", mo.fill_band_q4, "ksw", mo.min_ksw_len,
                  "bw", mo.bw, "gap", mo.max_gap, "skip", mo.chain_skip_q8, "margin", mo.fill_margin, "q2", mo.q2, "e2", mo.e2, "flags", mo.flags, "qtarget", qtarget is not None, "edge_repeats", edge_repeats)
            raise


if __name__ == "__main__":
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    run(Engine(0), n_iter, seed0, big=os.environ.get("FUZZ_BIG") is not None, sv=os.environ.get("FUZZ_SV") is not None,
        presets=os.environ["FUZZ_PRESETS"].split(",") if os.environ.get("FUZZ_PRESETS") else None, hard=os.environ.get("FUZZ_HARD") is not None)
    print("fuzz ok:", n_iter, "iterations in %.1f s" % (time.time() - t0))
