"""The packed form on the wire (telr_seqset_packed / telr_seqset_from_packed, round 4): the library's own device arrays handed
to torch as zero-copy tensors, and a set rebuilt from such words -- the two ends of the device-resident N > 1 hand-offs."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_packed_words_are_the_documented_layout_and_round_trip(engine, data_dir):
    import torch
    import packed_np
    from telr_amd.aligner import SeqSet
    from telr_amd.fasta import read_fasta
    from telr_amd.presets import preset
    _, ts = read_fasta(data_dir + "/ref_38kb.fasta")
    _, qs = read_fasta(data_dir + "/reads.fasta")
    qs = qs + ["", "ACGTNNNNACGT" * 7, "A"]
    s = engine.seqset(qs)
    w2, wn = s.packed()
    assert w2.is_cuda and w2.dtype == torch.int32 and len(w2) == 2 * len(wn)
    lens, e2, en = packed_np.pack(qs)                                  # the header's layout, restated with numpy
    np.testing.assert_array_equal(w2.cpu().numpy().view(np.uint32), e2)
    np.testing.assert_array_equal(wn.cpu().numpy().view(np.uint32), en)
    # a subset in another order, its words moved through torch (as an all-to-all would), a set rebuilt from them
    idx = np.array([5, 0, len(qs) - 2, 3, 3, len(qs) - 3], np.int32)
    sub = s.subset(idx)
    a, b = sub.packed()
    moved2, movedn = a.clone(), b.clone()
    sub.free()
    back = SeqSet.from_packed(engine, [len(qs[i]) for i in idx], moved2, movedn)
    assert back.n == len(idx) and back.bases() == sum(len(qs[i]) for i in idx)
    io, mo = preset("map-ont")
    ix = engine.index(ts, io)
    r1 = ix.map([qs[i] for i in idx], mo)
    raw = ix.map_raw(back, mo)
    n = engine.L.telr_result_count(raw)
    from telr_amd.aligner import _np_from
    from telr_amd._abi import ALN_DTYPE
    al = _np_from(engine.L.telr_result_alns(raw), n, ALN_DTYPE)
    assert n == len(r1.alns) > 0
    for f in ("qid", "tid", "qs", "qe", "ts", "te", "mlen", "blen", "dp_score", "flags", "mapq", "n_cigar"):
        np.testing.assert_array_equal(al[f], r1.alns[f], err_msg=f)
    ix.free_raw(raw); back.free(); s.free()
    with pytest.raises(Exception):                                     # word counts that do not fit the lengths are refused
        SeqSet.from_packed(engine, [100], moved2[:4], movedn[:2])


def test_reverse_complement_subset_equals_the_packed_reverse_complement(engine):
    """telr_seqset_subset_rc: the copies flagged rc are the reverse complements of their sources, word for word what the packer
    makes of the reverse-complemented strings (N stays N, lengths 0 / 1 / 63 / 64 / 65 / several words, repeats)"""
    import packed_np
    from telr_amd.fasta import revcomp
    rng = np.random.default_rng(5)
    seqs = ["", "A", "ACGTN", "".join(rng.choice(list("ACGT"), 63)), "".join(rng.choice(list("ACGT"), 64)), "".join(rng.choice(list("ACGTN"), 65)),
            "".join(rng.choice(list("ACGT"), 1000)), "N" * 40 + "".join(rng.choice(list("ACGT"), 333)) + "NN"]
    s = engine.seqset(seqs)
    idx = np.array([7, 0, 1, 2, 3, 3, 4, 5, 6, 6, 7], np.int32)
    rc = np.array([1, 1, 1, 1, 0, 1, 1, 1, 0, 1, 0], np.uint8)
    sub = s.subset(idx, rc=rc)
    w2, wn = sub.packed()
    want = [revcomp(seqs[i]) if f else seqs[i] for i, f in zip(idx, rc)]
    lens, e2, en = packed_np.pack(want)
    assert list(sub.len) == [len(x) for x in want]
    np.testing.assert_array_equal(w2.cpu().numpy().view(np.uint32), e2)
    np.testing.assert_array_equal(wn.cpu().numpy().view(np.uint32), en)
    with pytest.raises(ValueError):
        s.subset(idx, rc=rc[:3])
