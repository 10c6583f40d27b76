"""The provenance-tag cells of the packed DP (kernels.hip.h: d_cell_pk4 / d_cell_pk8) decide sources and extension flags by
carrying a tag in the low bits of scaled scores and taking plain maxima.  This restates both cells in Python integers and
checks, on random inputs with many ties, that the decisions equal the oracle's compare-and-select rule (band_dp in
oracle/telr_oracle.c: a gap extends only if strictly better than opening; H prefers diagonal > E1 > F1 > E2 > F2 on ties)."""
import random


def plain(hd, hl, e1l, e2l, hu, f1u, f2u, sc, q, e, q2, e2, two_piece):
    def gap(h, g, qq, ee):
        op, ex = h - (qq + ee), g - ee
        return max(op, ex), ex > op                     # extended only when strictly better
    E1, xE1 = gap(hl, e1l, q, e); F1, xF1 = gap(hu, f1u, q, e)
    cands = [hd + sc, E1, F1]
    flags = [xE1, xF1]
    if two_piece:
        E2, xE2 = gap(hl, e2l, q2, e2); F2, xF2 = gap(hu, f2u, q2, e2)
        cands += [E2, F2]; flags += [xE2, xF2]
    h = max(cands)
    src = cands.index(h)                                # first maximum = the oracle's preference order
    return h, src, flags, cands[1:]


def tagged(hd, hl, e1l, e2l, hu, f1u, f2u, sc, q, e, q2, e2, two_piece):
    S = 8 if two_piece else 4                           # scores times S, tags in the low log2(S) bits
    strip = lambda v: v & ~(S - 1)
    if not two_piece:
        et = max(S * hl - S * (q + e) + 2, (S * e1l | 1) - S * e)        # E states are stored under tag 1
        ft = max(S * hu - S * (q + e) + 1, S * f1u - S * e)              # F states under tag 0
        Es, Fs = strip(et) | 1, strip(ft)
        ht = max(S * hd + S * sc + 2, Es, Fs)
        src = 2 - (ht & 3)
        flags = [(et & 1) == 1, (ft & 1) == 0]
        states = [strip(Es) // S, Fs // S]
    else:
        e1 = max(S * hl - S * (q + e) + 4, (S * e1l | 3) - S * e)        # stored tags: E1 3, F1 2, E2 1, F2 0
        f1 = max(S * hu - S * (q + e) + 3, (S * f1u | 2) - S * e)
        e2_ = max(S * hl - S * (q2 + e2) + 2, (S * e2l | 1) - S * e2)
        f2 = max(S * hu - S * (q2 + e2) + 1, S * f2u - S * e2)
        E1s, F1s, E2s, F2s = strip(e1) | 3, strip(f1) | 2, strip(e2_) | 1, strip(f2)
        ht = max(S * hd + S * sc + 4, E1s, F1s, E2s, F2s)
        src = 4 - (ht & 7)
        src = {0: 0, 1: 1, 2: 2, 3: 3, 4: 4}[src]
        flags = [(e1 & 1) == 1, (f1 & 1) == 0, (e2_ & 1) == 1, (f2 & 1) == 0]
        states = [strip(E1s) // S, strip(F1s) // S, strip(E2s) // S, F2s // S]
    return strip(ht) // S, src, flags, states


def test_tagged_cells_decide_like_the_compare_and_select_rule():
    rng = random.Random(7)
    for two_piece in (False, True):
        for _ in range(20000):
            q, e = rng.randint(1, 8), rng.randint(1, 5)
            q2, e2 = q + rng.randint(0, 30), rng.randint(1, e)
            base = rng.randint(-300, 300)
            v = lambda: base + rng.randint(-6, 6)                      # close values: plenty of ties
            args = (v(), v(), v(), v(), v(), v(), v(), rng.choice((-4, -2, 1, 2)), q, e, q2, e2, two_piece)
            hp, sp, fp, stp = plain(*args)
            ht, st, ft, stt = tagged(*args)
            assert hp == ht and stp == stt, args
            assert sp == st, (args, sp, st)
            # a flag is only consulted for a state that lies on the path; here all of them are compared
            assert fp == ft, (args, fp, ft)
