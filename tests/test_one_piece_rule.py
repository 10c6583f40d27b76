"""The rule behind the one-piece variant of the packed DP cell (kernels.hip.h: d_cell_pk<ONEP>), checked on the CPU oracle:
in a band of D diagonals no gap run is longer than D - 1, so while (D - 1)(e - e2) < q2 - q the second affine piece
q2 + L e2 is strictly dearer than q + L e for every possible run, and switching it off (q2 = "infinity") changes neither
the score nor the CIGAR.  Beyond that width it can."""
import numpy as np

from oracle import binding as ob
from telr_amd import synth
from telr_amd.presets import preset


def _pair(rng, n, gap=0):
    t = synth.random_seq(rng, n)
    q = synth.mutate(rng, t, 0.05, 0.04, 0.04)
    if gap:
        q = np.concatenate([q[:n // 2], q[n // 2 + gap:]])            # one long deletion in the query = a long D run
    return bytes(q).decode(), bytes(t).decode()


def _band_diagonals(m, n, mo):
    mn = min(m, n)
    W = min(2 + ((mo.fill_band_q4 * int(np.floor(np.sqrt(mn)))) >> 4), mo.bw)
    lo = min(0, n - m) - W
    lo -= lo & 1
    return max(0, n - m) + W - lo + 1


def test_second_piece_cannot_pay_in_narrow_bands():
    rng = np.random.default_rng(17)
    for name in ("map-ont", "map-pb", "asm10", "ngmlr-pacbio"):
        _, mo = preset(name)
        limit = -(-(mo.q2 - mo.q) // (mo.e - mo.e2))                   # ceil: widest band of the rule
        one = mo.copy(); one.q2 = 30000; one.e2 = mo.e2
        checked = 0
        for it in range(300):
            q, t = _pair(rng, int(rng.integers(40, 400)))
            mo.fill_band_q4 = one.fill_band_q4 = int(rng.integers(1, 10))
            if _band_diagonals(len(q), len(t), mo) > limit:
                continue
            s2, c2 = ob.nw(q, t, mo)
            s1, c1 = ob.nw(q, t, one)
            assert s1 == s2 and c1.tolist() == c2.tolist(), (name, it)
            checked += 1
        assert checked >= 60, (name, checked)


def test_second_piece_does_pay_in_wide_bands():
    """the control: a 60-base deletion inside a band wide enough to hold it is cheaper through the second piece"""
    rng = np.random.default_rng(3)
    _, mo = preset("map-ont")
    mo.fill_band_q4 = 160                                               # W ~ 2 + 160 * 17 / 16: a band of hundreds of diagonals
    one = mo.copy(); one.q2 = 30000
    diff = 0
    for it in range(20):
        q, t = _pair(rng, 320, gap=60)
        s2, _ = ob.nw(q, t, mo)
        s1, _ = ob.nw(q, t, one)
        assert s2 >= s1
        diff += s2 > s1
    assert diff >= 15
