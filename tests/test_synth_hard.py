"""The HARD genome of telr_amd/synth.py (round 6: `bench.py --config c2r`, `tools/faithful_table.py --hard`, `FUZZ_HARD=1`): seeded,
deterministic, the sequence classes are really there, and the easy data sets are bit-identical to what they were (the hard options draw
from generators of their own)."""
import collections
import hashlib

import numpy as np

from telr_amd import synth


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_hard_features_and_determinism():
    kw = dict(seed=3, genome_len=2_000_000, n_reads=200, total_bases=1_800_000, n_ins=40)
    d1 = synth.make_stage1_dataset(hard=True, **kw)
    d2 = synth.make_stage1_dataset(hard=True, **kw)
    assert _sha(d1["ref"]) == _sha(d2["ref"]) and _sha(d1["reads"][0]) == _sha(d2["reads"][0])
    kinds = collections.Counter(f[2] for f in d1["hard_features"])
    bases = collections.Counter()
    for f in d1["hard_features"]:
        bases[f[2]] += f[1] - f[0]
    assert kinds["tandem"] >= 3 and kinds["micro"] >= 200 and kinds["lowcx"] >= 5 and kinds["segdup"] >= 1 and kinds["satellite"] >= 2, kinds
    assert 0.03 * 2e6 < bases["tandem"] < 0.10 * 2e6 and 10_000 <= bases["segdup"] <= 100_000, bases
    # a tandem array is periodic: its unit recurs; a satellite sits next to (not on) its insertion site
    # (copies are 0-5 % diverged WITH indels, so the phase drifts along an array and later features may overwrite parts of it: the
    # period is looked for in 400-base pieces, and most arrays must show one)
    def periodic(seg):
        return any((seg[p:] == seg[:-p]).mean() > 0.7 for p in range(2, min(201, len(seg) // 2)))
    arrays = [f for f in d1["hard_features"] if f[2] == "tandem" and f[1] - f[0] >= 1200]
    hits = sum(1 for s, e, _ in arrays if any(periodic(d1["ref"][x:x + 400]) for x in range(s, e - 400, 400)))
    assert arrays and hits >= 0.6 * len(arrays), (hits, len(arrays))
    for f in d1["hard_features"]:
        if f[2] == "satellite":
            site = f[3]
            assert f[1] <= site - 50 or f[0] >= site + 8 + 50, f


def test_easy_data_sets_are_unchanged_by_the_hard_options():
    kw = dict(seed=5, genome_len=400_000, n_reads=60, total_bases=300_000, n_ins=10)
    a = synth.make_stage1_dataset(**kw)
    b = synth.make_stage1_dataset(hard=None, **kw)
    assert _sha(a["ref"]) == _sha(b["ref"]) and _sha(a["reads"][0]) == _sha(b["reads"][0]) and a["hard_features"] == []
    g = synth.make_genome(7, [("a", 300_000), ("b", 120_000)], n_ins=8)
    gh = synth.make_genome(7, [("a", 300_000), ("b", 120_000)], n_ins=8, hard=synth.HARD)
    assert g["insertions"] == gh["insertions"] and g["hard_features"] == [] and len(gh["hard_features"]) > 20
    assert (g["ref"][0] != gh["ref"][0]).mean() > 0.02
    plan = synth.plan_reads(g, 1.0)
    r0 = synth.materialize_reads(g, plan)
    r1 = synth.materialize_reads(g, plan, burst=None)
    rb = synth.materialize_reads(g, plan, burst=synth.HARD["burst"])
    assert _sha(r0[0]) == _sha(r1[0]) and _sha(r0[0]) != _sha(rb[0])


def test_error_bursts_raise_the_local_error_rate():
    rng = np.random.default_rng(1)
    seq = synth.random_seq_fast(rng, 400_000)
    out, ln = synth.mutate_bulk(np.random.default_rng(2), seq, np.array([len(seq)]), 0.0, 0.0, 0.04, burst=(1 / 5000.0, 200, 200, 5.0))
    # deletions only: 4 % outside the bursts, 20 % inside; bursts cover ~4 % of the bases -> ~4.6 % overall
    lost = 1.0 - ln[0] / len(seq)
    assert 0.043 < lost < 0.052, lost
