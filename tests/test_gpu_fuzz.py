"""Randomised parity: random genomes, read sets (tiny / empty / N-containing reads included) and option mixes (k, w, look-back,
band factor and retry margin, segment length, band width, gap limit, two-piece gap costs on both sides of the one-piece rule, skip penalty,
secondary output, per-target ranking, extension limits, all five presets),
every stage compared bit for bit with the oracle.  `python tests/fuzz_parity.py 800 <seed>` runs the long version."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_configurations(engine, seed):
    import fuzz_parity
    fuzz_parity.run(engine, 40, seed)


@pytest.mark.parametrize("seed", [21, 22])
def test_random_configurations_with_large_indels_under_the_convex_cost(engine, seed):
    """reads with 30-600-base insertions / deletions on the `ngmlr-*` presets: fills in bands of 128-1,024 diagonals and long gap
    runs, i.e. the re-biased int16 classes of the convex cost (one wave, several waves) and the int32 classes behind them"""
    import fuzz_parity
    fuzz_parity.run(engine, 12, seed, sv=True, presets=["ngmlr-ont", "ngmlr-pacbio"])


@pytest.mark.parametrize("seed,big", [(31, False), (32, False), (33, True)])
def test_random_configurations_on_hard_sequence(engine, seed, big):
    """round 6: targets with tandem arrays, microsatellites, low-complexity stretches and (Mb-size targets) segmental duplications,
    reads with error bursts -- the sequence classes of `bench.py --config c2r`: queries past the LDS sort's limit (the library's
    segmented sort for those only), hundreds of equal chains, extensions through arrays; every stage against the oracle"""
    import fuzz_parity
    fuzz_parity.run(engine, 4 if big else 30, seed, big=big, hard=True)
