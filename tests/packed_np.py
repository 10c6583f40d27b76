"""TEST HELPER: the packed form of a sequence set (include/telr_hip.h, telr_seqset_packed) made and read with numpy, so that the
CPU (gloo) tests of the device-resident exchanges can fabricate and check the word tensors without a GPU.  The product never
packs on the host this way: it hands the library's own device arrays to the collectives."""
import numpy as np

_CODE = np.full(256, 4, np.uint8)
for _i, _c in enumerate("ACGT"):
    _CODE[ord(_c)] = _i; _CODE[ord(_c.lower())] = _i
_CODE[ord("U")] = 3; _CODE[ord("u")] = 3


def pack(seqs):
    """list of str / uint8 arrays -> (lengths int32, 2-bit words uint32, mask words uint32)"""
    lens = np.array([len(s) for s in seqs], np.int32)
    tot = int((((lens.astype(np.int64) + 63) // 64) * 64).sum())
    code = np.zeros(tot, np.uint8); amb = np.zeros(tot, np.uint8)
    o = 0
    for s, L in zip(seqs, lens):
        a = np.frombuffer(s.encode(), np.uint8) if isinstance(s, str) else np.asarray(s, np.uint8)
        c = _CODE[a]
        code[o:o + L] = np.where(c < 4, c, 0); amb[o:o + L] = c >= 4
        o += ((int(L) + 63) // 64) * 64
    w2 = (code.reshape(-1, 16).astype(np.uint32) << (2 * np.arange(16, dtype=np.uint32))).sum(axis=1, dtype=np.uint32)
    wn = (amb.reshape(-1, 32).astype(np.uint32) << np.arange(32, dtype=np.uint32)).sum(axis=1, dtype=np.uint32)
    return lens, w2, wn


def unpack(lens, w2, wn):
    """-> list of str (A C G T, N where the mask bit is set)"""
    w2 = np.asarray(w2).view(np.uint32); wn = np.asarray(wn).view(np.uint32)
    code = ((w2[:, None] >> (2 * np.arange(16, dtype=np.uint32))) & 3).astype(np.uint8).reshape(-1)
    amb = ((wn[:, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(np.uint8).reshape(-1)
    out, o = [], 0
    lut = np.frombuffer(b"ACGT", np.uint8)
    for L in lens:
        L = int(L)
        b = lut[code[o:o + L]].copy(); b[amb[o:o + L] == 1] = ord("N")
        out.append(b.tobytes().decode())
        o += ((L + 63) // 64) * 64
    return out


def subset_words(lens, w2, wn, idx):
    """the packed words of sequences idx (in that order): what telr_seqset_subset + telr_seqset_packed give on the device"""
    blocks = (np.asarray(lens, np.int64) + 63) // 64
    start = np.concatenate([[0], np.cumsum(blocks)])
    p2 = [w2[start[i] * 4:(start[i] + blocks[i]) * 4] for i in idx]
    pn = [wn[start[i] * 2:(start[i] + blocks[i]) * 2] for i in idx]
    z = np.zeros(0, np.uint32)
    return (np.concatenate(p2) if p2 else z), (np.concatenate(pn) if pn else z)
