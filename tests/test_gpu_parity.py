"""HIP engine vs the CPU oracle, stage by stage and end to end (bit-exact: integer work)."""
import numpy as np
import pytest

from telr_amd.presets import preset
from telr_amd.fasta import read_fasta, concat
from telr_amd import synth

pytestmark = pytest.mark.gpu

ALN_FIELDS = ["qid", "tid", "qlen", "qs", "qe", "tlen", "ts", "te", "mlen", "blen", "score", "subsc", "dp_score", "cnt",
              "n_sub", "parent", "n_cigar", "flags", "mapq"]


def _as_str(a):
    return bytes(a).decode() if not isinstance(a, str) else a


def compare_all(engine, targets, queries, io, mo, qtarget=None, stages=True):
    from oracle import binding as ob
    targets = [_as_str(t) for t in targets]
    queries = [_as_str(q) for q in queries]
    oix = ob.OracleIndex(targets, io)
    gix = engine.index(targets, io)
    # --- index
    oh, oy = oix.dump()
    eh, eo, pos = gix.debug_dump()
    assert len(pos) == len(oy), "minimizer count: gpu %d oracle %d" % (len(pos), len(oy))
    np.testing.assert_array_equal(pos, oy)
    np.testing.assert_array_equal(np.repeat(eh, np.diff(eo.astype(np.int64))), oh)
    assert gix.debug_mid_occ(mo) == oix.mid_occ(mo)
    # --- map
    oref = oix.map(queries, mo, qtarget=qtarget, debug=True)
    res = gix.map(queries, mo, qtarget=qtarget)
    if stages:
        dbg = gix.debug_last_batch(len(queries))
        np.testing.assert_array_equal(dbg["q_aoff"].astype(np.int64), oref["anchor_off"])
        np.testing.assert_array_equal(dbg["skeys"], oref["anchors"])
        np.testing.assert_array_equal(dbg["chain_f"], oref["f"])
        np.testing.assert_array_equal(dbg["chain_p"], oref["p"])
        np.testing.assert_array_equal(dbg["chains"], oref["chains"])
    assert len(res.alns) == len(oref["alns"])
    for f in ALN_FIELDS:
        np.testing.assert_array_equal(res.alns[f], oref["alns"][f], err_msg="field " + f)
    for i in range(len(res.alns)):
        a, b = res.alns[i], oref["alns"][i]
        np.testing.assert_array_equal(res.cigars[a["cigar_off"]:a["cigar_off"] + a["n_cigar"]],
                                      oref["cigars"][b["cigar_off"]:b["cigar_off"] + b["n_cigar"]], err_msg="cigar of record %d" % i)
    ctr = engine.counters()
    for k in ("query_bases", "minimizers", "anchors", "chains", "dp_problems", "dp_cells", "window_bases", "records", "cigar_ops"):
        assert ctr[k] == oref["counters"][k], (k, ctr[k], oref["counters"][k])
    return res, oref


def test_fixture_map_ont(engine, data_dir):
    _, ts = read_fasta(data_dir + "/ref_38kb.fasta")
    _, qs = read_fasta(data_dir + "/reads.fasta")
    io, mo = preset("map-ont")
    res, _ = compare_all(engine, ts, qs, io, mo)
    assert len(res.alns) >= 18


def test_fixture_library_to_reads_asm10(engine, data_dir):
    """TE library (jockey) against the bundled reads as targets: many targets, asm10 scoring."""
    _, lib = read_fasta(data_dir + "/library.fasta")
    _, qs = read_fasta(data_dir + "/reads.fasta")
    io, mo = preset("asm10")
    mo.best_n = 10
    compare_all(engine, qs, lib, io, mo)


@pytest.mark.parametrize("seed,lookback", [(1, 64), (2, 128), (3, 256)])
def test_synthetic_reads(engine, seed, lookback):
    rng = np.random.default_rng(20261002 + seed)
    genome = [synth.random_seq(rng, 200000), synth.random_seq(rng, 60000)]
    # a repeat family so that the occurrence filter and secondary chains are exercised
    te = synth.random_seq(rng, 3000)
    for g in genome:
        for _ in range(6):
            p = int(rng.integers(0, len(g) - 3000))
            g[p:p + 3000] = synth.mutate(rng, te, 0.03, 0.0, 0.0)[:3000]
    reads, truth = synth.simulate_reads(rng, genome, 60, 6000)
    # one read with ambiguous bases, one shorter than k, one empty
    reads[3][100:130] = ord("N")
    reads.append(np.frombuffer(b"ACGTACG", dtype=np.uint8))
    reads.append(np.zeros(0, np.uint8))
    io, mo = preset("map-ont")
    mo.chain_lookback = lookback
    res, _ = compare_all(engine, genome, reads, io, mo)
    # truth recovery: primary alignments land on the simulated origin
    prim = res.alns[(res.alns["flags"] & 1) != 0]
    ok = 0
    for a in prim:
        if a["qid"] >= len(truth):
            continue
        g, s, e, st = truth[a["qid"]]
        if a["tid"] == g and a["ts"] < e and a["te"] > s and ((a["flags"] >> 3) & 1) == st:
            ok += 1
    assert ok >= 0.9 * len(truth)


@pytest.mark.parametrize("lookback,skip_q8,bw,max_gap", [(64, 3, 500, 5000), (128, 7, 500, 5000), (256, 2, 300, 800), (128, 0, 6000, 2000)])
def test_chain_options(engine, lookback, skip_q8, bw, max_gap):
    """the chaining kernel's second variant (non-zero skip penalty, not used by any preset) and band / gap settings on both
    sides of the folded range test (bw < max_gap, bw >= max_gap): f / p arrays, chains and records equal to the oracle"""
    rng = np.random.default_rng(20261002 + 50 + lookback + skip_q8)
    genome = [synth.random_seq(rng, 150000)]
    te = synth.random_seq(rng, 2000)
    for _ in range(8):
        p = int(rng.integers(0, len(genome[0]) - 2000))
        genome[0][p:p + 2000] = synth.mutate(rng, te, 0.04, 0.0, 0.0)[:2000]
    reads, _ = synth.simulate_reads(rng, genome, 40, 5000)
    io, mo = preset("map-ont")
    mo.chain_lookback = lookback; mo.chain_skip_q8 = skip_q8; mo.bw = bw; mo.max_gap = max_gap
    compare_all(engine, genome, reads, io, mo)


def test_target_filter_and_per_target(engine):
    """S3/S4/S6 shape: query i sees only target qtarget[i]; S5 shape: per-target selection."""
    rng = np.random.default_rng(7)
    contigs = [synth.random_seq(rng, 20000) for _ in range(5)]
    te = synth.random_seq(rng, 2500)
    for c in contigs:
        c[9000:11500] = synth.mutate(rng, te, 0.02, 0.0, 0.0)[:2500]
    reads, qt = [], []
    for ci, c in enumerate(contigs):
        for _ in range(6):
            s = int(rng.integers(0, 12000)); L = int(rng.integers(3000, 8000))
            r = c[s:s + L]
            if rng.integers(0, 2):
                r = synth.revcomp_arr(r)
            reads.append(synth.mutate(rng, r)); qt.append(ci)
    io, mo = preset("map-ont")
    res, _ = compare_all(engine, contigs, reads, io, mo, qtarget=np.array(qt, np.int32))
    assert (res.alns["tid"] == np.array(qt)[res.alns["qid"]]).all()
    mo2 = mo.copy(); mo2.flags |= 2
    res2, _ = compare_all(engine, contigs, [te], io, mo2)
    assert set(res2.alns["tid"].tolist()) == set(range(5))


def test_depth_medians(engine):
    from oracle import binding as ob
    rng = np.random.default_rng(11)
    contig = synth.random_seq(rng, 15000)
    reads = []
    for _ in range(40):
        s = int(rng.integers(0, 9000)); L = int(rng.integers(2000, 6000))
        reads.append(synth.mutate(rng, contig[s:s + L]))
    io, mo = preset("map-ont")
    gix = engine.index([bytes(contig).decode()], io)
    r = gix.map_raw([bytes(x).decode() for x in reads], mo)
    try:
        res = gix.result_arrays(r)
        iv_s = np.array([0, 5000, 7000, 14950, 100], np.int32); iv_e = np.array([50, 5100, 7050, 15100, 14000], np.int32)
        iv_t = np.zeros(5, np.int32)
        got = gix.depth_medians(r, iv_t, iv_s, iv_e)
        want = ob.depth_medians(res.alns, res.cigars, [15000], iv_t, iv_s, iv_e)
        np.testing.assert_array_equal(got, want)
        assert got[1] > 3
    finally:
        gix.free_raw(r)


def test_wide_bands(engine):
    """Anchor-free stretches force long gap-fill segments: register kernel with 2/4/8 diagonal pairs
    per lane, the LDS kernel and (asm10, bw=10000) very wide bands."""
    rng = np.random.default_rng(99)
    genome = synth.random_seq(rng, 120000)
    reads = []
    for gap in (1500, 3000, 4500, 900, 2200):
        for rep in range(2):
            s = int(rng.integers(0, 100000)); L = 12000 + gap
            r = genome[s:s + L].copy()
            a = 5000
            r[a:a + gap] = synth.mutate(rng, r[a:a + gap], 0.35, 0.0, 0.0)       # no shared minimizers in here
            if rep:
                # unequal lengths: delete part of the read inside the stretch
                r = np.concatenate([r[:a + 100], r[a + 100 + gap // 8:]])
            r = synth.mutate(rng, r, 0.02, 0.01, 0.01)
            if rng.integers(0, 2):
                r = synth.revcomp_arr(r)
            reads.append(r)
    io, mo = preset("map-ont")
    res, oref = compare_all(engine, [genome], reads, io, mo)
    assert oref["counters"]["dp_problems"] > 0
    io2, mo2 = preset("asm10")
    compare_all(engine, [genome], reads, io2, mo2)


def test_adaptive_band_retry(engine):
    """A 20-base insertion compensated 60 bases later by a 20-base deletion leaves the segment's end points on
    one diagonal but drags the optimal path outside the narrow first-pass band: the narrow pass touches the
    band edge and the problem is re-aligned with the wide band (same rule in the oracle)."""
    rng = np.random.default_rng(123)
    genome = synth.random_seq(rng, 60000)
    reads = []
    for k in range(12):
        s = int(rng.integers(0, 40000)); r = genome[s:s + 9000].copy()
        parts, pos = [], 0
        for x in range(1500, 8000, 1500):
            ins = synth.random_seq(rng, 20 + k % 5)
            parts += [r[pos:x], ins, r[x:x + 60]]
            pos = x + 60 + 20 + k % 5                  # delete the same number of bases 60 bp downstream
        parts.append(r[pos:])
        r = synth.mutate(rng, np.concatenate(parts), 0.01, 0.003, 0.003)
        if k % 2:
            r = synth.revcomp_arr(r)
        reads.append(r)
    io, mo = preset("map-ont")
    res, oref = compare_all(engine, [genome], reads, io, mo)
    assert int(engine.L.telr_debug_dp_retries(engine.h)) > 0
    prim = res.alns[(res.alns["flags"] & 1) != 0]
    assert len(prim) == len(reads) and (prim["qe"] - prim["qs"] > 8500).all()


def test_fixture_map_pb_hpc(engine, data_dir):
    """The reference's own smoke configuration: PacBio reads, `--presets pacbio` -> map-pb (homopolymer-compressed
    k-mers, k=19): device HPC compaction + sketch against the oracle."""
    _, ts = read_fasta(data_dir + "/ref_38kb.fasta")
    _, qs = read_fasta(data_dir + "/reads.fasta")
    io, mo = preset("map-pb")
    res, _ = compare_all(engine, ts, qs, io, mo)
    assert len(res.alns) >= 18


def test_synthetic_clr_hpc(engine):
    rng = np.random.default_rng(31)
    genome = [synth.random_seq(rng, 150000), synth.random_seq(rng, 50)]
    genome[0][5000:5400] = ord("A")                       # a 400-base homopolymer: span >= 256 k-mers are skipped
    genome[0][70000:70020] = ord("N")
    reads, truth = synth.simulate_reads(rng, genome[:1], 40, 7000, err=(0.013, 0.065, 0.052))
    reads.append(np.frombuffer(b"ACGTTTTTTTTTTTTTTTTTTTTTTTTTTTTGCA", dtype=np.uint8))   # fewer than k runs
    reads.append(np.zeros(0, np.uint8))
    io, mo = preset("map-pb")
    res, _ = compare_all(engine, genome, reads, io, mo)
    prim = res.alns[(res.alns["flags"] & 1) != 0]
    assert len(prim) >= 38


def _revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTNacgtn", "TGCANtgcan"))


def test_sam_and_paf_emitters(engine, data_dir, tmp_path):
    """SAM records must be self-consistent: SEQ + CIGAR + MD reproduce the reference bases, cs agrees with MD,
    NM = mismatches + gap bases, clip lengths add up; PAF columns equal the record fields."""
    import re
    tn, ts = read_fasta(data_dir + "/ref_38kb.fasta")
    qn, qs = read_fasta(data_dir + "/reads.fasta")
    io, mo = preset("map-ont")
    ix = engine.index(ts, io)
    r = ix.map_raw(qs + ["ACGTACGTAACC" * 3], mo)
    try:
        res = ix.result_arrays(r)
        sam, paf = str(tmp_path / "o.sam"), str(tmp_path / "o.paf")
        names = qn + ["unmapped_read"]
        ix.write_sam(r, names, qs + ["ACGTACGTAACC" * 3], tn, ts, sam, rg=("s1", "s1", "ont"))
        ix.write_paf(r, names, tn, paf)
    finally:
        ix.free_raw(r)
    recs = [l.rstrip("\n").split("\t") for l in open(sam) if not l.startswith("@")]
    hdr = [l for l in open(sam) if l.startswith("@")]
    assert any(l.startswith("@SQ\tSN:%s\tLN:%d" % (tn[0], len(ts[0]))) for l in hdr) and any(l.startswith("@RG\tID:s1") for l in hdr)
    assert len(recs) == len(res.alns) + 1 and recs[-1][1] == "4"
    ref = ts[0].upper()
    n_checked = 0
    for f, a in zip(recs, res.alns):
        flag, pos, cigar, seq = int(f[1]), int(f[3]) - 1, f[5], f[9]
        tags = {t[:2]: t[5:] for t in f[11:]}
        assert pos == a["ts"] and f[2] == tn[a["tid"]] and int(f[4]) == a["mapq"]
        assert bool(flag & 0x10) == bool(a["flags"] & 8) and bool(flag & 0x100) == bool(a["flags"] & 2) and bool(flag & 0x800) == bool(a["flags"] & 4)
        ops = re.findall(r"(\d+)([MIDSH])", cigar)
        qlen_c = sum(int(n) for n, o in ops if o in "MIS")
        if seq != "*":
            assert qlen_c == len(seq)
        assert sum(int(n) for n, o in ops if o in "MISH") == a["qlen"]
        assert sum(int(n) for n, o in ops if o in "MD") == a["te"] - a["ts"]
        assert int(tags["AS"]) == a["dp_score"] and tags["RG"] == "s1"
        if seq == "*":
            continue
        # rebuild the reference from SEQ + CIGAR + MD
        qi = 0; aligned_q = []
        for n, o in ops:
            n = int(n)
            if o == "S":
                qi += n
            elif o == "M":
                aligned_q.append(seq[qi:qi + n]); qi += n
            elif o == "I":
                qi += n
        mseq = "".join(aligned_q).upper()
        out, mi = [], 0
        for m in re.finditer(r"(\d+)|(\^[A-Z]+)|([A-Z])", tags["MD"]):
            if m.group(1):
                k = int(m.group(1)); out.append(mseq[mi:mi + k]); mi += k
            elif m.group(2):
                out.append(m.group(2)[1:])
            else:
                out.append(m.group(3)); mi += 1
        assert "".join(out) == ref[a["ts"]:a["te"]], "MD does not rebuild the reference for " + f[0]
        nm = sum(int(n) for n, o in ops if o in "ID") + len(re.findall(r"(?<![\^A-Z])[A-Z]", re.sub(r"\^[A-Z]+", "^", tags["MD"])))
        assert int(tags["NM"]) == nm == a["blen"] - a["mlen"]
        # cs: total reference / query lengths
        cs_t = sum(int(x) for x in re.findall(r":(\d+)", tags["cs"])) + len(re.findall(r"\*", tags["cs"])) + sum(len(x) for x in re.findall(r"-([a-z]+)", tags["cs"]))
        assert cs_t == a["te"] - a["ts"]
        n_checked += 1
    assert n_checked >= 18
    # supplementary records carry SA tags pointing at each other
    assert any("SA" in {t[:2] for t in f[11:]} for f in recs[:-1])
    plines = [l.rstrip("\n").split("\t") for l in open(paf)]
    assert len(plines) == len(res.alns)
    for p, a in zip(plines, res.alns):
        assert [int(p[i]) for i in (1, 2, 3, 6, 7, 8, 9, 10, 11)] == [a[k] for k in ("qlen", "qs", "qe", "tlen", "ts", "te", "mlen", "blen", "mapq")]
        assert p[4] == ("-" if a["flags"] & 8 else "+") and p[5] == tn[a["tid"]]


def _read_bgzf(path):
    import struct, zlib
    data = open(path, "rb").read()
    out, off, blocks = [], 0, []
    while off < len(data):
        assert data[off:off + 4] == b"\x1f\x8b\x08\x04"
        xlen = struct.unpack_from("<H", data, off + 10)[0]
        assert data[off + 12:off + 14] == b"BC"
        bsize = struct.unpack_from("<H", data, off + 16)[0] + 1
        raw = zlib.decompress(data[off + 12 + xlen:off + bsize - 8], -15)
        assert struct.unpack_from("<I", data, off + bsize - 4)[0] == len(raw)
        assert struct.unpack_from("<I", data, off + bsize - 8)[0] == zlib.crc32(raw)
        blocks.append((off, len(b"".join(out)), len(raw)))
        out.append(raw); off += bsize
    assert blocks[-1][2] == 0          # EOF marker block
    return b"".join(out), blocks


def test_sorted_bam_and_bai(engine, data_dir, tmp_path):
    """telr_alignment.alignment() == map + `samtools sort` + `samtools index`: BGZF/BAM/BAI parsed back here."""
    import struct
    from telr_amd import telr_alignment
    bam = str(tmp_path / "s_sort.bam")
    telr_alignment.alignment(bam, data_dir + "/reads.fasta", data_dir + "/ref_38kb.fasta", str(tmp_path), "s", 1, "minimap2", "ont", engine=engine)
    raw, blocks = _read_bgzf(bam)
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].decode()
    assert "SO:coordinate" in text and "@SQ\tSN:" in text
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]; p += 4
    assert n_ref == 1
    l_name = struct.unpack_from("<i", raw, p)[0]; p += 4 + l_name
    l_ref = struct.unpack_from("<i", raw, p)[0]; p += 4
    assert l_ref == 38001
    recs, last = [], (-1, -1)
    while p < len(raw):
        bs, refid, pos, lrn, mapq, bn, ncig, flag, lseq = struct.unpack_from("<iiiBBHHHi", raw, p)
        start = p; p += 4
        body = raw[p:p + bs]; p += bs
        name = body[32:32 + lrn - 1].decode()
        cig = struct.unpack_from("<%dI" % ncig, body, 32 + lrn)
        reflen = sum(c >> 4 for c in cig if (c & 0xf) in (0, 2))
        qlen_c = sum(c >> 4 for c in cig if (c & 0xf) in (0, 1, 4))
        if refid >= 0:
            assert (refid, pos) >= last; last = (refid, pos)
            assert flag & 0x100 or qlen_c == lseq
            # bin as the SAM spec defines it
            beg, end = pos, pos + reflen - 1
            want = 0
            for sh, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
                if beg >> sh == end >> sh:
                    want = base + (beg >> sh); break
            assert bn == want
        recs.append((start, refid, pos, name, flag, reflen, bn))
    mapped = [r for r in recs if r[1] >= 0]
    assert len(mapped) >= 18 and all(r[1] == -1 for r in recs[len(mapped):])
    # the same records as the engine produced
    io, mo = preset("map-ont")
    _, ts = read_fasta(data_dir + "/ref_38kb.fasta"); _, qs = read_fasta(data_dir + "/reads.fasta")
    res = engine.index(ts, io).map(qs, mo)
    assert sorted(int(a["ts"]) for a in res.alns) == [r[2] for r in mapped]
    assert sorted(int(a["te"] - a["ts"]) for a in res.alns) == sorted(r[5] for r in mapped)
    # BAI: every record start lies inside a chunk of its bin; linear index is monotone
    bai = open(bam + ".bai", "rb").read()
    assert bai[:4] == b"BAI\x01" and struct.unpack_from("<i", bai, 4)[0] == 1
    q = 8
    n_bin = struct.unpack_from("<i", bai, q)[0]; q += 4
    bins = {}
    for _ in range(n_bin):
        b, nch = struct.unpack_from("<Ii", bai, q); q += 8
        bins[b] = [struct.unpack_from("<QQ", bai, q + 16 * c) for c in range(nch)]; q += 16 * nch
    n_intv = struct.unpack_from("<i", bai, q)[0]; q += 4
    lin = struct.unpack_from("<%dQ" % n_intv, bai, q); q += 8 * n_intv
    assert list(lin) == sorted(lin) and n_intv == (38000 >> 14) + 1
    ustart = {b[1]: b[0] for b in blocks}

    def voff(u):
        blk = max(k for k in ustart if k <= u)
        return ustart[blk] << 16 | (u - blk)
    for start, refid, pos, name, flag, reflen, bn in mapped:
        v = voff(start)
        assert any(c0 <= v < c1 for c0, c1 in bins[bn]), "record at %d not covered by bin %d" % (pos, bn)
        assert lin[pos >> 14] <= v
    assert 37450 in bins and bins[37450][1][0] == len(mapped)


def test_pipelined_ranges(engine):
    """Two ranges of a call are in flight at a time (second slot: a context of its own) and append to the one result in range
    order.  Forced here on a small input, with an odd and an even number of ranges and with one range only: records and CIGARs
    must equal the oracle's single-batch result."""
    import os
    rng = np.random.default_rng(4242)
    genome = [synth.random_seq(rng, 120000), synth.random_seq(rng, 50000)]
    reads, _ = synth.simulate_reads(rng, genome, 60, 3500)
    long_reads, _ = synth.simulate_reads(rng, genome, 5, 25000)
    reads[7:7] = long_reads[:3]; reads.extend(long_reads[3:]); reads.insert(20, np.zeros(0, np.uint8))
    io, mo = preset("map-ont")
    qt = np.array([i % 3 - 1 for i in range(len(reads))], np.int32)
    for kbp in ("60", "45", "1000"):
        os.environ["TELR_PIPELINE"] = "force"; os.environ["TELR_BATCH_KBP"] = kbp
        try:
            res, _ = compare_all(engine, genome, reads, io, mo, stages=False)
            compare_all(engine, genome, reads, io, mo, qtarget=qt, stages=False)
        finally:
            del os.environ["TELR_PIPELINE"], os.environ["TELR_BATCH_KBP"]
        assert len(res.alns) >= 60 and (np.diff(res.alns["qid"]) >= 0).all()
    # the fall-back when two ranges in flight do not fit the device: the second slot is released and the call runs again one
    # range at a time (forced here after a successful pipelined attempt)
    os.environ["TELR_PIPELINE"] = "force"; os.environ["TELR_BATCH_KBP"] = "60"; os.environ["TELR_TEST_PIPE_NOMEM"] = "1"
    try:
        res, _ = compare_all(engine, genome, reads, io, mo, stages=False)
        res2, _ = compare_all(engine, genome, reads, io, mo, qtarget=qt, stages=False)
    finally:
        del os.environ["TELR_PIPELINE"], os.environ["TELR_BATCH_KBP"], os.environ["TELR_TEST_PIPE_NOMEM"]
    assert len(res.alns) >= 60 and (np.diff(res.alns["qid"]) >= 0).all()


def test_per_query_target_call_runs_one_range_and_falls_back_to_ranges_in_turn(engine):
    """round 6: a call with per-query targets (S6, the polishing map) that the range plan would cut runs as ONE range (what it waits
    for is its longest chaining run, once per range), and as the plan's ranges in turn when the device has no room for that (forced
    here after a successful attempt): the oracle's records either way.  4,200 short queries: the plan applies from 4,000 on."""
    import os
    rng = np.random.default_rng(4343)
    genome = [synth.random_seq(rng, 60000), synth.random_seq(rng, 40000), synth.random_seq(rng, 30000)]
    reads, truth = synth.simulate_reads(rng, genome, 4200, 400)
    qt = np.array([t[0] for t in truth], np.int32)
    io, mo = preset("map-ont")
    os.environ["TELR_BATCH_KBP"] = "500"
    try:
        res, _ = compare_all(engine, genome, reads, io, mo, qtarget=qt, stages=False)
        os.environ["TELR_TEST_PIPE_NOMEM"] = "1"
        res2, _ = compare_all(engine, genome, reads, io, mo, qtarget=qt, stages=False)
    finally:
        del os.environ["TELR_BATCH_KBP"]; os.environ.pop("TELR_TEST_PIPE_NOMEM", None)
    assert len(res.alns) >= 3000 and (np.diff(res.alns["qid"]) >= 0).all() and len(res2.alns) == len(res.alns)


def test_cigars_kept_on_the_device_equal_the_host_array(engine):
    """TELR_MF_KEEP_CIGARS: the device copy of a result's CIGAR array (for telr_write_bam_dev) is the host array, word for word --
    one range, pipelined ranges with and without the long-read lanes (whose pieces are merged in on the host and uploaded),
    after the two-slot fall-back, and with per-query targets; and the host arrays are what they are without the flag"""
    import os
    from telr_amd._abi import MF_KEEP_CIGARS
    rng = np.random.default_rng(99)
    genome = [synth.random_seq(rng, 120000), synth.random_seq(rng, 50000)]
    reads, _ = synth.simulate_reads(rng, genome, 60, 3500)
    long_reads, _ = synth.simulate_reads(rng, genome, 5, 25000)
    reads[7:7] = long_reads[:3]; reads.extend(long_reads[3:]); reads.insert(20, np.zeros(0, np.uint8))
    io, mo = preset("map-ont")
    mk = type(mo).from_buffer_copy(mo); mk.flags |= MF_KEEP_CIGARS
    ix = engine.index(concat(genome), io); qs = engine.seqset(concat(reads))
    qt = np.array([i % 3 - 1 for i in range(len(reads))], np.int32)

    def check(qtarget=None):
        r0 = ix.map_raw(qs, mo, qtarget=qtarget); r1 = ix.map_raw(qs, mk, qtarget=qtarget)
        try:
            a, b = ix.result_arrays(r0), ix.result_arrays(r1)
            assert a.alns.tobytes() == b.alns.tobytes() and a.cigars.tobytes() == b.cigars.tobytes() and len(b.cigars) > 1000
            dev = np.zeros(len(b.cigars), np.uint32)
            assert engine.L.telr_debug_result_twin(r1, dev.ctypes.data, len(dev)) == len(dev)
            assert (dev == b.cigars).all()
            assert engine.L.telr_debug_result_twin(r0, dev.ctypes.data, len(dev)) == -1          # not asked for: no device copy
        finally:
            ix.free_raw(r0); ix.free_raw(r1)
    check(); check(qt)
    for env in ({"TELR_PIPELINE": "force", "TELR_BATCH_KBP": "60"}, {"TELR_PIPELINE": "force", "TELR_BATCH_KBP": "45"},
                {"TELR_PIPELINE": "force", "TELR_BATCH_KBP": "1000"}, {"TELR_PIPELINE": "force", "TELR_BATCH_KBP": "60", "TELR_TEST_PIPE_NOMEM": "1"}):
        os.environ.update(env)
        try:
            check(); check(qt)
        finally:
            for k in env:
                del os.environ[k]


def test_seqset_subset_matches_fresh_set(engine):
    """telr_seqset_subset gathers packed sequences on the device: mapping the subset must give exactly what mapping a
    freshly packed set of the same sequences gives (repeats, an empty read and the last read included)."""
    rng = np.random.default_rng(77)
    genome = [synth.random_seq(rng, 90000)]
    reads, _ = synth.simulate_reads(rng, genome, 30, 4000)
    reads.insert(5, np.zeros(0, np.uint8))
    io, mo = preset("map-ont")
    ix = engine.index([_as_str(g) for g in genome], io)
    parent = engine.seqset([_as_str(r) for r in reads])
    idx = [len(reads) - 1, 3, 3, 5, 0, 17]
    sub = parent.subset(idx)
    assert sub.n == len(idx) and sub.bases() == sum(len(reads[i]) for i in idx)
    a = ix.map(sub, mo)
    b = ix.map([_as_str(reads[i]) for i in idx], mo)
    assert len(a.alns) == len(b.alns) > 0
    for f in ALN_FIELDS:
        np.testing.assert_array_equal(a.alns[f], b.alns[f], err_msg=f)
    for x, y in zip(a.alns, b.alns):
        np.testing.assert_array_equal(a.cigars[x["cigar_off"]:x["cigar_off"] + x["n_cigar"]], b.cigars[y["cigar_off"]:y["cigar_off"] + y["n_cigar"]])
    with pytest.raises(Exception):
        parent.subset([len(reads)])


def test_edge_cases(engine):
    """empty / degenerate inputs and argument errors at the C ABI (no crash, empty result is not an error)"""
    from telr_amd._lib import TelrError
    rng = np.random.default_rng(8)
    g = bytes(synth.random_seq(rng, 5000)).decode()
    io, mo = preset("map-ont")
    ix = engine.index([g, "ACGT", ""], io)                       # targets shorter than k and empty
    assert len(ix.map([], mo).alns) == 0                         # no queries
    res = ix.map(["", "ACG", "N" * 500, g[1000:1014]], mo)       # empty, shorter than k, all ambiguous, shorter than k+w
    assert len(res.alns) == 0
    res = ix.map([g[500:3500]], mo)
    assert len(res.alns) == 1 and res.alns[0]["tid"] == 0 and res.alns[0]["mlen"] == 3000 and res.cigar_string(0) == "3000M"
    # compare the degenerate index against the oracle as well
    compare_all(engine, [g, "ACGT", ""], ["", "ACG", "N" * 500, g[500:3500], g[100:160]], io, mo)
    with pytest.raises(TelrError):
        ix.map([g[:1000]], mo, qtarget=np.array([7], np.int32))  # target id out of range
    bad = mo.copy(); bad.chain_lookback = 100
    with pytest.raises(TelrError):
        ix.map([g[:1000]], bad)
    bad = mo.copy(); bad.e, bad.e2 = 1, 2
    with pytest.raises(TelrError):
        ix.map([g[:1000]], bad)
    empty_ix = engine.index([], io)
    assert len(empty_ix.map([g[:1000]], mo).alns) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("pname", ["map-ont", "map-pb"])
def test_long_join_reads_across_large_insertions_and_deletions(engine, pname):
    """spec 3.11 (minimap2 -r500,20000): a read across a multi-kb insertion or deletion is ONE chain and one record whose CIGAR
    carries the long gap; the fill across it is the two-band DP (DP class 0, problem kind 5).  HIP = oracle at every stage."""
    rng = np.random.default_rng(4242)
    genome = [synth.random_seq(rng, 400000), synth.random_seq(rng, 150000)]
    err = (0.013, 0.065, 0.052) if pname == "map-pb" else (0.04, 0.02, 0.04)
    reads, want = [], []
    for k in range(60):
        g = genome[k % 2]
        p = int(rng.integers(10000, len(g) - 30000))
        left = g[p:p + int(rng.integers(1500, 6000))]
        kind = k % 3
        if kind == 0:          # insertion of 600 .. 4500 bases
            L = int(rng.integers(600, 4500))
            r = np.concatenate([left, synth.random_seq(rng, L), g[p + len(left):p + len(left) + int(rng.integers(1500, 6000))]])
        elif kind == 1:        # deletion of 600 .. 4500 bases
            L = int(rng.integers(600, 4500))
            r = np.concatenate([left, g[p + len(left) + L:p + len(left) + L + int(rng.integers(1500, 6000))]])
        else:                  # a tandem duplication of the junction (TSD-like) around an insertion
            L = int(rng.integers(700, 3000)); tsd = int(rng.integers(4, 9))
            r = np.concatenate([left, synth.random_seq(rng, L), g[p + len(left) - tsd:p + len(left) + int(rng.integers(1500, 5000))]])
        r = synth.mutate(rng, r, *err)
        if rng.integers(0, 2):
            r = synth.revcomp_arr(r)
        reads.append(r); want.append((kind, L))
    io, mo = preset(pname)
    assert mo.bw_long == 20000
    res, oref = compare_all(engine, genome, reads, io, mo)
    big = 0
    for i, a in enumerate(res.alns):
        ops = res.cigar(i)
        if any((c & 0xf) in (1, 2) and (c >> 4) >= 500 for c in ops):
            big += 1
            kind, L = want[a["qid"]]
            gaps = [int(c >> 4) for c in ops if (c & 0xf) == (2 if kind == 1 else 1) and (c >> 4) >= 500]
            assert gaps and abs(max(gaps) - L) <= 60 + 0.08 * L, (want[a["qid"]], gaps)          # L is counted before the read errors (+-6 %)
    assert big >= 40, big                      # most of the 60 reads are one record with the long gap inside
    # without the long join the same reads split (what the ngmlr presets do by design)
    off = mo.copy(); off.bw_long = 0
    res0, _ = compare_all(engine, genome, reads, io, off, stages=False)
    assert len(res0.alns) > len(res.alns)


def test_second_context_and_scratch_release(engine):
    """Engine.worker() (telr_init_background: a context of its own at the lowest stream priority) maps a read set of the first
    context against an index of the first context to the same records; telr_release_scratch gives the grow-only scratch back and
    the next call sizes it again with the same result; telr_device_mem reports the difference"""
    rng = np.random.default_rng(17)
    genome = [synth.random_seq(rng, 200000)]
    reads, _ = synth.simulate_reads(rng, genome, 80, 4000)
    io, mo = preset("map-ont")
    ix = engine.index(concat(genome), io); qs = engine.seqset(concat(reads))
    a = ix.map(qs, mo)
    w = engine.worker()
    assert w is engine.worker() and w.h.value != engine.h.value
    import ctypes as C
    h = C.c_void_p()
    w._chk(w.L.telr_map(w.h, ix.h, qs.h, None, C.byref(mo), C.byref(h)), "telr_map on the second context")
    try:
        b = ix.result_arrays(h)
        assert a.alns.tobytes() == b.alns.tobytes() and a.cigars.tobytes() == b.cigars.tobytes()
    finally:
        ix.free_raw(h)
    free0, total = engine.mem_info()
    engine.release_scratch(); w.release_scratch()
    free1, _ = engine.mem_info()
    assert free1 > free0 and total > free1
    c = ix.map(qs, mo)
    assert a.alns.tobytes() == c.alns.tobytes() and a.cigars.tobytes() == c.cigars.tobytes()


def test_hpc_chain_start_at_the_target_start(engine):
    """engine == oracle where the query's homopolymer runs are longer than the target's at the very start of the target (the chain
    start clamps at 0: tests/test_oracle.py has the construction)"""
    from test_oracle import _hpc_start_case
    targets, reads = _hpc_start_case()
    io, mo = preset("map-pb")
    res, _ = compare_all(engine, targets, reads, io, mo, stages=False)
    assert (res.alns["ts"] >= 0).all() and (res.alns["ts"] <= 5).any()
    compare_all(engine, targets + targets, reads, io, mo, qtarget=np.array([1, 0, 1], np.int32), stages=False)
