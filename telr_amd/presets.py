"""Option presets for the seven aligner call sites of the reference.

`-x map-ont` / `map-pb` are what TELR_alignment.py:57-60, TELR_assembly.py:193-197 and
TELR_te.py:595-598 derive from `--presets {ont,pacbio}`; `asm10` is hard-wired in
TELR_te.py:899 -> TELR_liftover.py:254-264.  Parameter values follow the minimap2 2.22
manual as recorded in SURVEY.md 8(a) [recall]; `telr_preset()` in the C library returns
the same numbers (tests/test_abi.py keeps them in lock-step).
"""
import threading

from ._abi import IdxOpt, MapOpt, MF_CIGAR


def _gap_q8(k, scale=0.8):
    # chain gap cost 0.01 * scale * k per base of diagonal drift, as Q8 fixed point
    return int(0.01 * scale * k * 256 + 0.5)


_OVERRIDES = {}          # field -> value laid over every preset while `override(...)` is active (experiments, A/B legs of bench.py)
_OV_LOCK = threading.Lock()      # preset() is called from the S6 worker thread too: the table is swapped, never edited in place


class override:
    """`with presets.override(bw_long=0): ...` -- every preset() call inside (any thread) gets these map-option fields on top of
    the table's.  For experiments only: e.g. bench.py reports the per-locus call set with and without the long join at S4-S6."""
    def __init__(self, **fields):
        self.fields = fields

    def __enter__(self):
        global _OVERRIDES
        with _OV_LOCK:
            self.saved = _OVERRIDES
            _OVERRIDES = dict(self.saved, **self.fields)         # a NEW dict: a reader holds either the old table or the new one
        return self

    def __exit__(self, *exc):
        global _OVERRIDES
        with _OV_LOCK:
            _OVERRIDES = self.saved
        return False


def preset(name):
    io = IdxOpt(k=15, w=10, is_hpc=0, bucket_bits=0)
    mo = MapOpt(
        mid_occ_frac=2e-4, min_mid_occ=10, max_mid_occ=1000000,
        max_gap=5000, bw=500, chain_lookback=128, min_cnt=3, min_chain_score=40,
        chain_gap_q8=0, chain_skip_q8=0,
        mask_level=0.5, pri_ratio=0.8, best_n=5, secondary=1,
        a=2, b=4, q=4, e=2, q2=24, e2=1, sc_ambi=1, zdrop=400, min_dp_max=80, min_ksw_len=200,
        ext_max=2048, ext_band=31, flags=MF_CIGAR, fill_band_q4=6, fill_margin=1,
        vote_len=0, vote_bin_shift=0, vote_min=0, vote_frac_q8=0, bw_long=0,
        cx_scale=0, cx_open=0, cx_ext_max=0, cx_ext_min=0, cx_decay=0)
    if name in ("map-ont", "map-pb"):
        mo.bw_long = 20000          # minimap2 -r500,20000: a read across a multi-kb insertion / deletion is one chain
    if name == "map-ont":
        mo.fill_band_q4 = 4         # with fill_margin 1: no record of the faithful-mode gate differs (tests/test_faithful_gate.py; DESIGN.md, band rule)
    elif name == "map-pb":
        io.k, io.is_hpc = 19, 1
        mo.fill_band_q4 = 12        # real CLR reads (the bundled fixture) have bursty indels: 8 still loses two of its 25 records to the full band
        # round 6, the HARD genome (profiles/r06_faithful_table_hard.md): the 2,048-base cap of the end extensions moved 0.70 % of 4,701 records'
        # coordinates (a read whose last kilobases lie in a tandem array has no anchors there: the extension has to cross them); 3,900 bases
        # -- still inside the packed int16 extension class, m + n <= 7,854 -- leaves what the +-31 band moves
        mo.ext_max = 3900
    elif name in ("ngmlr-ont", "ngmlr-pacbio"):
        # `ngmlr -x ont|pacbio`, the reference's default stage-1 aligner (TELR_alignment.py:28-51, TELR_input.py:176-177):
        # 13-mers at every third reference position = (w,k) = (5,13) minimizers; NGMLR's convex gap cost as the lower
        # envelope of two affine pieces (derivation in telr_engine.hip: telr_preset)
        io.k, io.w = 13, 5
        if name == "ngmlr-ont":
            mo.a, mo.b, mo.q, mo.e, mo.q2, mo.e2 = 2, 2, 2, 2, 4, 1
            # round 4: NGMLR's convex gap cost in exact form (length-tracking cells, scores in 1/10 of this preset's unit): the faithful
            # gate measured 3.2 % of the records' coordinates against the two-piece envelope above, so the exact form is the spec
            mo.cx_scale, mo.cx_open, mo.cx_ext_max, mo.cx_ext_min, mo.cx_decay = 10, 20, 20, 10, 3
        else:
            mo.a, mo.b, mo.q, mo.e, mo.q2, mo.e2 = 2, 5, 6, 4, 60, 1
            # exact as well (scores in 1/20: ext(i) = max(20, 100 - 3 i)); the envelope moved 0.18 % of the coordinates
            mo.cx_scale, mo.cx_open, mo.cx_ext_max, mo.cx_ext_min, mo.cx_decay = 20, 100, 100, 20, 3
        # NGMLR's candidate search: 256-base sub-reads vote for reference regions (diagonal bins of 32 bases, a window of three
        # bins = its corridor), regions with at least half the votes of the sub-read's best one stay
        mo.vote_len, mo.vote_bin_shift, mo.vote_min, mo.vote_frac_q8 = 256, 5, 3, 128
        # round 6, the HARD genome (tandem arrays, satellites, segmental duplications, reads with error bursts; profiles/r06_faithful_table_hard.md):
        # with 13-mers a sub-read brings hundreds of repeat hits, they sort between the true anchors, and a look-back of 128 anchors no longer
        # reaches across them: against look-back 5,000 it moved 1.1 % of ~4,900 records (1.0-1.4 % of the coordinates) on both presets -- over
        # the 0.5 % rule -- and broke chains whose pieces then ran into the extension cap (0.9 % of the coordinates).  256: 0.29 % / 0.37 % (ont),
        # 0.26 % / 0.11 % (pacbio), and the extension rows fall to 0.22 % / 0.37 % and 0.00 % / 0.15 % with it.
        mo.chain_lookback = 256
        mo.fill_band_q4, mo.fill_margin = 12, 2         # cheap gaps let paths wander: the band the faithful-mode gate needs on the fixture
        if name == "ngmlr-ont":
            # round 5, both measured on the oracle (tests/test_faithful_gate.py, profiles/r05_faithful_table.md):
            # (1) fills: factor 7 with a retry margin of 4 diagonals gives the full-band result on every record of the fixture and of the
            #     ONT gate sample with 18 % fewer cells than (12, 2) (CLR reads: (12, 2) stays the cheapest setting without drift);
            # (2) end extensions in +-63 diagonals: under this preset's cheap gaps an extension into non-homologous sequence (a read
            #     clipped at an insertion) keeps gaining a little and wanders; against extensions without band or length cap +-31 moved
            #     1.1-1.4 % of the records' coordinates (the round-4 rule violation), +-63 moves 0.46 %.
            mo.fill_band_q4, mo.fill_margin = 7, 4
            mo.ext_band = 63
            # (3) z-drop 100 (minimap2's 400 belongs to ITS scores; NGMLR has none): 9 % of this preset's extensions run into non-homologous
            #     sequence (the clipped side of a split read), where its scores never fall 400 below their maximum and the DP ran on to ext_max
            #     = 2,048 bases: two thirds of all extension cells, for a best cell found long before.  With 100 (a dip of 50 of NGMLR's
            #     mismatches, or a 96-base indel) every record of the 1,080-record gate sample and of the fixture is unchanged (down to 50;
            #     25 changes one fixture record of ngmlr-pacbio), the DP computes 7.5 % (sample) / 24 % (fixture) fewer cells, and the row of
            #     the gate drops to 0.28 % / 0.00 %.
            mo.zdrop = 100
    elif name == "asm10":
        io.k, io.w = 19, 19
        mo.min_mid_occ, mo.max_mid_occ = 50, 500
        mo.bw, mo.max_gap = 10000, 10000
        mo.a, mo.b, mo.q, mo.e, mo.q2, mo.e2 = 1, 9, 16, 2, 41, 1
        mo.min_dp_max, mo.zdrop, mo.best_n = 200, 200, 50
    else:
        raise ValueError("unknown preset %r" % (name,))
    mo.chain_gap_q8 = _gap_q8(io.k)
    for f, v in _OVERRIDES.items():          # (one reference read: see override)
        setattr(mo, f, v)
    return io, mo
