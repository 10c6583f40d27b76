"""The device-side BAM writer (telr_write_bam_dev: records, coordinate sort, BGZF blocks and CRC-32 made by kernels) against
the host writer (telr_write_bam: the CIGAR walk over ASCII sequences on host threads, zlib): after inflating, the two files
must hold the SAME BYTES -- header, every record (fixed fields, CIGAR with clips, 4-bit SEQ, QUAL, NM / AS / MD / cs / SA /
tp / cm / s1 / s2 / RG tags), in the same coordinate order -- and the .bai must index the device file's own blocks
(reference hand-off H1: src/telr/TELR_alignment.py:103-114)."""
import struct
import zlib

import numpy as np
import pytest

from telr_amd import synth
from telr_amd.fasta import read_fasta, concat
from telr_amd.presets import preset
from test_gpu_parity import _read_bgzf

pytestmark = pytest.mark.gpu


def _records(raw):
    """[(offset, refid, pos, flag, name, n_cigar, l_seq, body)] of an inflated BAM stream + offset of the first record"""
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]; p += 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", raw, p)[0]; p += 4 + ln + 4
    first = p
    out = []
    while p < len(raw):
        bs, refid, pos, lrn, mapq, bn, ncig, flag, lseq = struct.unpack_from("<iiiBBHHHi", raw, p)
        body = raw[p + 4:p + 4 + bs]
        out.append((p, refid, pos, flag, body[32:32 + lrn - 1].decode(), ncig, lseq, body))
        p += 4 + bs
    assert p == len(raw)
    return out, first


def _check_bai(bam, raw, blocks, n_ref):
    recs, _ = _records(raw)
    bai = open(bam + ".bai", "rb").read()
    assert bai[:4] == b"BAI\x01" and struct.unpack_from("<i", bai, 4)[0] == n_ref
    ustart = sorted((b[1], b[0]) for b in blocks)          # (uncompressed start, file offset)
    us = np.array([u for u, _ in ustart]); fo = [f for _, f in ustart]

    def voff(u):
        k = int(np.searchsorted(us, u, side="right")) - 1
        return fo[k] << 16 | (u - int(us[k]))
    q = 8
    per_ref = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", bai, q)[0]; q += 4
        bins = {}
        for _ in range(n_bin):
            b, nch = struct.unpack_from("<Ii", bai, q); q += 8
            bins[b] = [struct.unpack_from("<QQ", bai, q + 16 * c) for c in range(nch)]; q += 16 * nch
        n_intv = struct.unpack_from("<i", bai, q)[0]; q += 4
        lin = struct.unpack_from("<%dQ" % n_intv, bai, q); q += 8 * n_intv
        assert list(lin) == sorted(lin)
        per_ref.append((bins, lin))
    n_no_coor = struct.unpack_from("<Q", bai, q)[0]; q += 8
    assert q == len(bai)
    assert n_no_coor == sum(1 for r in recs if r[1] < 0)
    cnt = [0] * n_ref
    for off, refid, pos, flag, name, ncig, lseq, body in recs:
        if refid < 0:
            continue
        cnt[refid] += 1
        bn = struct.unpack_from("<H", body, 10)[0]
        bins, lin = per_ref[refid]
        v = voff(off)
        assert any(c0 <= v < c1 for c0, c1 in bins[bn]), "record at %d not covered by a chunk of bin %d" % (pos, bn)
        assert lin[pos >> 14] <= v
    for t in range(n_ref):
        if cnt[t]:
            assert per_ref[t][0][37450][1][0] == cnt[t]


def _both(engine, ts, tnames, qs, qnames, pname, tmp_path, tag, rg=None, softclip=True, md=True, cs=True):
    io, mo = preset(pname)
    ix = engine.index(ts, io)
    qset = engine.seqset(qs)
    r = ix.map_raw(qset, mo)
    try:
        h, d = str(tmp_path / (tag + "_host.bam")), str(tmp_path / (tag + "_dev.bam"))
        ix.write_bam(r, qnames, qs, tnames, ts, h, md=md, cs=cs, softclip=softclip, rg=rg, cmdline="t", index=True, level=1)
        ix.write_bam_device(r, qset, qnames, tnames, d, md=md, cs=cs, softclip=softclip, rg=rg, cmdline="t", index=True, level=0)
        z = str(tmp_path / (tag + "_devz.bam"))
        ix.write_bam_device(r, qset, qnames, tnames, z, md=md, cs=cs, softclip=softclip, rg=rg, cmdline="t", index=True, level=1)
    finally:
        ix.free_raw(r)
    rh, bh = _read_bgzf(h)
    rd, bd = _read_bgzf(d)          # inflates every block and checks its CRC-32 and ISIZE
    if rh != rd:
        a, _ = _records(rh); b, _ = _records(rd)
        assert len(a) == len(b), (len(a), len(b))
        for x, y in zip(a, b):
            assert x[1:7] == y[1:7], (x[:7], y[:7])
            if x[7] != y[7]:
                k = next(i for i in range(min(len(x[7]), len(y[7]))) if x[7][i] != y[7][i])
                raise AssertionError("record %s at %d: bodies differ at byte %d of %d / %d: %r vs %r" % (x[4], x[2], k, len(x[7]), len(y[7]), x[7][max(0, k - 24):k + 24], y[7][max(0, k - 24):k + 24]))
        raise AssertionError("streams differ outside the records")
    _check_bai(d, rd, bd, len(tnames))
    # level 1: deflate blocks coded on the device (per-field Huffman tables, run-length matches): zlib inflates them to the same stream
    import os
    rz, bz = _read_bgzf(z)
    assert rz == rh
    _check_bai(z, rz, bz, len(tnames))
    if len(rh) > 200000:
        assert os.path.getsize(z) < 0.5 * os.path.getsize(d), (os.path.getsize(z), os.path.getsize(d), os.path.getsize(h))
    _both.sizes = (os.path.getsize(h), os.path.getsize(d), os.path.getsize(z))
    return _records(rd)[0]


def test_fixture_device_bam_equals_host_bam(engine, data_dir, tmp_path):
    tn, ts = read_fasta(data_dir + "/ref_38kb.fasta")
    qn, qs = read_fasta(data_dir + "/reads.fasta")
    recs = _both(engine, ts, tn, qs, qn, "map-ont", tmp_path, "fx")
    assert len(recs) >= 18
    recs = _both(engine, ts, tn, qs, qn, "ngmlr-pacbio", tmp_path, "fxrg", rg=("s1", "s1", "pb"), cs=False)
    assert all(b"RGZs1\x00" in r[7] for r in recs)
    _both(engine, ts, tn, qs, qn, "map-pb", tmp_path, "fxhard", softclip=False)


def test_synthetic_device_bam_equals_host_bam(engine, tmp_path):
    """several targets, both strands, supplementary + secondary records, unmapped and empty reads, reads with N, a read longer
    than a BGZF block, lower-case and IUPAC bases (printed as N by both writers)"""
    rng = np.random.default_rng(77)
    genome = [synth.random_seq(rng, 300000), synth.random_seq(rng, 120000), synth.random_seq(rng, 50000)]
    # a repeat so that secondaries appear, an N run inside the first target
    genome[1][20000:26000] = genome[0][100000:106000]
    genome[0][150000:150400] = ord("N")
    reads, _ = synth.simulate_reads(rng, genome, 160, 6000)
    # a chimeric read (supplementary record), a read of random bases (unmapped), an empty read, a read with Ns and IUPAC codes
    a = synth.mutate(rng, genome[0][5000:13000])
    b = genome[2][10000:19000].copy()
    reads.append(np.concatenate([a, synth.revcomp_arr(b)]))
    reads.append(synth.random_seq(rng, 3000))
    reads.append(np.zeros(0, np.uint8))
    x = genome[0][200000:207000].copy(); x[100:103] = ord("N"); x[2000] = ord("R"); x[2500:2600] |= 32
    reads.append(x)
    reads.append(genome[0][20000:140000].copy())          # 120 kb: record spans two BGZF blocks
    names = ["r%d/x" % i for i in range(len(reads))]
    tn = ["chrA", "chrB_long_name", "c3"]
    recs = _both(engine, genome, tn, reads, names, "map-ont", tmp_path, "syn")
    flags = [r[3] for r in recs]
    assert any(f & 0x800 for f in flags) and any(f & 0x100 for f in flags) and any(f & 0x10 for f in flags) and any(f == 4 for f in flags)
    assert sum(1 for r in recs if r[1] < 0) >= 2
    _both(engine, genome, tn, reads, names, "ngmlr-ont", tmp_path, "synrg", rg=("smp", "smp", "ont"))
    _both(engine, genome, tn, reads, names, "map-ont", tmp_path, "synplain", md=False, cs=False)


def test_long_cigar_and_nothing_mapped(engine, tmp_path):
    """(a) a 520-kb read with a dense error pattern: more than 65,535 CIGAR operations, so the record carries the -L placeholder
    (<l_seq>S<ref_len>N) and the real CIGAR in CG:B,I -- in both writers, byte for byte; (b) a read set of which nothing maps:
    unmapped records only (refID -1, bin 4680, flag 4), and the same with TELR_SAM_NO_UNMAPPED: header only."""
    rng = np.random.default_rng(9)
    g = synth.random_seq(rng, 600000)
    long_read = synth.mutate(rng, g[20000:540000], 0.05, 0.04, 0.05)
    reads = [long_read, synth.mutate(rng, g[1000:9000], 0.03, 0.01, 0.01)]
    recs = _both(engine, [g], ["big"], reads, ["long/1", "short"], "map-ont", tmp_path, "lc")
    r0 = [r for r in recs if r[4] == "long/1" and not r[3] & 0x900][0]
    assert r0[5] == 2 and b"CGBI" in r0[7]                               # two placeholder ops in the record, the CIGAR in the tag
    ncg = struct.unpack_from("<I", r0[7], r0[7].index(b"CGBI") + 4)[0]
    assert ncg > 65535
    junk = [synth.random_seq(rng, 2000) for _ in range(7)] + [np.zeros(0, np.uint8)]
    recs = _both(engine, [g], ["big"], junk, ["j%d" % i for i in range(8)], "map-ont", tmp_path, "um")
    assert len(recs) == 8 and all(r[1] == -1 and r[3] == 4 for r in recs)
    io, mo = preset("map-ont")
    ix = engine.index([g], io); qset = engine.seqset(junk)
    r = ix.map_raw(qset, mo)
    try:
        p = str(tmp_path / "none.bam")
        ix.write_bam_device(r, qset, ["j%d" % i for i in range(8)], ["big"], p, unmapped=False, level=1)
    finally:
        ix.free_raw(r)
    raw, _ = _read_bgzf(p)
    assert _records(raw)[0] == []


def _map_synthetic(engine, n_reads=400, seed=5):
    rng = np.random.default_rng(seed)
    genome = [synth.random_seq(rng, 400000), synth.random_seq(rng, 150000)]
    reads, _ = synth.simulate_reads(rng, genome, n_reads, 7000)
    reads.append(synth.random_seq(rng, 2500))                      # an unmapped read
    names = ["q%d" % i for i in range(len(reads))]
    io, mo = preset("map-ont")
    ix = engine.index(concat(genome), io)
    qset = engine.seqset(concat(reads))
    return ix, qset, names, ["tA", "tB"], mo


def test_prepared_output_file_gives_the_same_bam(engine, tmp_path):
    """telr_bam_prepare (the file allocated, mapped and pre-faulted in the background, cut to its length at the end) with an
    estimate that is generous, far too small (the rest goes through pwrite), for ANOTHER path (ignored), twice in a row
    (the first mapping still being taken apart), and never used: always the file the unprepared writer makes, byte for byte"""
    import os
    ix, qset, names, tn, mo = _map_synthetic(engine)
    r = ix.map_raw(qset, mo)
    try:
        plain = str(tmp_path / "plain.bam")
        ix.write_bam_device(r, qset, names, tn, plain, cmdline="t", level=1)
        want = open(plain, "rb").read(); want_bai = open(plain + ".bai", "rb").read()
        assert len(want) > (1 << 20)
        for tag, est, prep_path in (("big", 300 << 20, None), ("exact", len(want), None), ("small", 200 << 10, None), ("tiny", 1, None), ("other", 8 << 20, "elsewhere.bam"),
                                    ("again", 100 << 20, None)):
            out = str(tmp_path / (tag + ".bam"))
            ix.bam_prepare(str(tmp_path / prep_path) if prep_path else out, est)
            ix.write_bam_device(r, qset, names, tn, out, cmdline="t", level=1)
            assert os.path.getsize(out) == len(want), (tag, os.path.getsize(out), len(want))
            assert open(out, "rb").read() == want and open(out + ".bai", "rb").read() == want_bai, tag
        # TELR_MF_KEEP_CIGARS: the result keeps its CIGAR array on the device too and the writer reads it there -- same records,
        # same CIGARs on the host, same file
        from telr_amd._abi import MF_KEEP_CIGARS
        mo2 = type(mo).from_buffer_copy(mo); mo2.flags |= MF_KEEP_CIGARS
        r2 = ix.map_raw(qset, mo2)
        try:
            a, b = ix.result_arrays(r), ix.result_arrays(r2)
            assert a.alns.tobytes() == b.alns.tobytes() and a.cigars.tobytes() == b.cigars.tobytes()
            kept = str(tmp_path / "kept.bam")
            ix.write_bam_device(r2, qset, names, tn, kept, cmdline="t", level=1)
            assert engine.L.telr_debug_bam_twin() == 1
            assert open(kept, "rb").read() == want and open(kept + ".bai", "rb").read() == want_bai
            ix.write_bam_device(r, qset, names, tn, kept, cmdline="t", level=1)
            assert engine.L.telr_debug_bam_twin() == 0 and open(kept, "rb").read() == want
        finally:
            ix.free_raw(r2)
        ix.bam_prepare(str(tmp_path / "unused.bam"), 50 << 20)      # dropped by the next prepare / by telr_destroy
        ix.bam_prepare(str(tmp_path / "unused2.bam"), 50 << 20)
        ix.bam_release_wait()
        lvl0 = str(tmp_path / "l0.bam")
        ix.bam_prepare(lvl0, 64 << 20)
        ix.write_bam_device(r, qset, names, tn, lvl0, cmdline="t", level=0)
        assert _read_bgzf(lvl0)[0] == _read_bgzf(plain)[0]
    finally:
        ix.free_raw(r)


@pytest.mark.parametrize("switch", ["bam_no_populate", "bam_no_twin"])
def test_writer_switches_in_a_process_of_their_own(tmp_path, switch):
    """TELR_AB=bam_no_populate (the prepared file's blocks allocated, not pre-faulted: the round's earlier sink) and
    TELR_AB=bam_no_twin (the writer uploads the CIGARs although the result kept them on the device): read once per process,
    so each runs in its own; the file must be the one the default path writes"""
    import os, subprocess, sys
    code = r"""
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, %r)
from telr_amd.aligner import Engine
from telr_amd._abi import MF_KEEP_CIGARS
import test_gpu_bam_dev as T
eng = Engine(0)
ix, qset, names, tn, mo = T._map_synthetic(eng, 200, 9)
mk = type(mo).from_buffer_copy(mo); mk.flags |= MF_KEEP_CIGARS
r = ix.map_raw(qset, mo); rk = ix.map_raw(qset, mk)
a, b = sys.argv[1] + "/a.bam", sys.argv[1] + "/b.bam"
ix.write_bam_device(r, qset, names, tn, a, cmdline="t", level=1)
ix.bam_prepare(b, 40 << 20)
ix.write_bam_device(rk, qset, names, tn, b, cmdline="t", level=1)
assert eng.L.telr_debug_bam_twin() == (0 if "bam_no_twin" in os.environ.get("TELR_AB", "") else 1)
assert open(a, "rb").read() == open(b, "rb").read() and open(a + ".bai", "rb").read() == open(b + ".bai", "rb").read()
assert eng.L.telr_bam_release_wait() == 0
print("same")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code, str(tmp_path)], env=dict(os.environ, TELR_AB=switch), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0 and b"same" in p.stdout, p.stderr.decode()[-2000:]
