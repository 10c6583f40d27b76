#!/usr/bin/env python3
"""Walk a whole BAM the slow, independent way (zlib + struct, no library of this repository): every BGZF block inflates, its
CRC-32 and ISIZE hold, the blocks' payloads concatenate to a BAM stream whose records chain by their block_size to the last byte,
records are in coordinate order (refID, pos; unmapped last), and the .bai's linear-index offsets are non-decreasing and point
at record starts.  Prints one JSON line.  usage: validate_bam.py file.bam [expected number of records]"""
import json, struct, sys, zlib

path = sys.argv[1]
raw = open(path, "rb").read()
p = 0; n_blk = 0; chunks = []; blk_off = []; u_off = []; u = 0
while p < len(raw):
    assert raw[p:p + 4] == b"\x1f\x8b\x08\x04", ("not a BGZF block", p)
    xlen = struct.unpack_from("<H", raw, p + 10)[0]
    assert raw[p + 12:p + 16] == b"BC\x02\x00"
    bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
    data = zlib.decompress(raw[p + 12 + xlen:p + bsize - 8], -15)
    crc, isize = struct.unpack_from("<II", raw, p + bsize - 8)
    assert zlib.crc32(data) == crc and len(data) == isize, ("crc / isize", p)
    blk_off.append(p); u_off.append(u); u += len(data)
    chunks.append(data); p += bsize; n_blk += 1
assert len(chunks[-1]) == 0, "no EOF block"
s = b"".join(chunks); del chunks
assert s[:4] == b"BAM\x01"
l_text = struct.unpack_from("<i", s, 4)[0]; q = 8 + l_text
n_ref = struct.unpack_from("<i", s, q)[0]; q += 4
for _ in range(n_ref):
    ln = struct.unpack_from("<i", s, q)[0]; q += 4 + ln + 4
n_rec = 0; last = (-1, -1); n_unmapped = 0; rec_starts = set(); seen_unmapped = False; bases = 0
while q < len(s):
    bs, refid, pos, lrn, mapq, bn, ncig, flag, lseq = struct.unpack_from("<iiiBBHHHi", s, q)
    rec_starts.add(q)
    if refid < 0:
        seen_unmapped = True; n_unmapped += 1
    else:
        assert not seen_unmapped, "mapped record after an unmapped one"
        assert (refid, pos) >= last, ("not sorted", n_rec, last, (refid, pos))
        last = (refid, pos)
    assert 32 + lrn + 4 * ncig + (lseq + 1) // 2 + lseq <= bs, ("fields exceed block_size", n_rec)
    bases += lseq
    q += 4 + bs; n_rec += 1
assert q == len(s), "records do not end at the end of the stream"
out = {"file": path, "bytes": len(raw), "bgzf_blocks": n_blk, "uncompressed_bytes": len(s), "references": n_ref, "records": n_rec, "unmapped": n_unmapped, "seq_bases": bases}
try:
    bai = open(path + ".bai", "rb").read()
    assert bai[:4] == b"BAI\x01" and struct.unpack_from("<i", bai, 4)[0] == n_ref
    import bisect
    b = 8; n_lin = 0; n_chunks = 0
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", bai, b)[0]; b += 4
        for _ in range(n_bin):
            bin_id, nch = struct.unpack_from("<Ii", bai, b)
            if bin_id != 37450:                      # every chunk of a real bin begins at a record start (the pseudo-bin holds offsets and counts)
                for c in range(nch):
                    v = struct.unpack_from("<Q", bai, b + 8 + 16 * c)[0]
                    k = bisect.bisect_right(blk_off, v >> 16) - 1
                    assert blk_off[k] == v >> 16 and (u_off[k] + (v & 0xffff)) in rec_starts, "chunk does not begin at a record start"
                n_chunks += nch
            b += 8 + 16 * nch
        n_intv = struct.unpack_from("<i", bai, b)[0]; b += 4
        lin = struct.unpack_from("<%dQ" % n_intv, bai, b); b += 8 * n_intv
        prev = 0
        for v in lin:
            assert v >= prev; prev = v
            if v:
                k = bisect.bisect_right(blk_off, v >> 16) - 1
                assert blk_off[k] == v >> 16 and (u_off[k] + (v & 0xffff)) in rec_starts, "linear index entry is not a record start"
        n_lin += n_intv
    out["bai_linear_entries"] = n_lin; out["bai_chunks"] = n_chunks
except FileNotFoundError:
    out["bai_linear_entries"] = None
if len(sys.argv) > 2:
    assert n_rec == int(sys.argv[2]), (n_rec, sys.argv[2])
print(json.dumps(out))
