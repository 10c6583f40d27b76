"""In-process interval arithmetic standing in for the bedtools 2.30 calls on the hot path.

Reference call sites (src/telr/TELR_liftover.py): `bedtools sort` :244, `closest -s -d -t all`
:501-518, `closest -d -D ref -k 5` :306-324, `getfasta` :167,208, `sort` + `merge -d 0 -c 4 -o
collapse`-style de-duplication :1108-1116.  bedtools itself is not in this image; semantics are
restated from its documentation [recall, SURVEY.md A.7]:
  * coordinates are 0-based half-open; features overlap when they share at least one base;
  * `closest -d` distance is 0 for overlapping features, otherwise (gap in bases + 1), so
    book-ended features are at distance 1;
  * `-t all` reports every tie; with no candidate the B columns are `.`/`-1` and distance -1;
  * `-D ref` signs the distance: negative when B lies upstream (lower coordinates) of A;
  * `-k N` reports the N closest (ties of the N-th included);
  * `sort` orders by chromosome (byte order) then start (stable).
A BED row is a list of strings exactly as it would appear in the file.
"""


def _null_b(ncol):
    """bedtools' filler for "no B feature": '.' for text columns, -1 for start / end / score"""
    return [".", "-1", "-1", ".", "-1", "."][:ncol] + ["."] * max(0, ncol - 6)


def bed_sort(rows):
    return sorted(rows, key=lambda r: (r[0], int(r[1])))


def _dist(a_s, a_e, b_s, b_e):
    """unsigned bedtools distance and the side of B relative to A (-1 upstream, +1 downstream, 0 overlap)"""
    if b_e <= a_s:
        return a_s - b_e + 1, -1
    if b_s >= a_e:
        return b_s - a_e + 1, 1
    return 0, 0


def closest_same_strand(a_rows, b_rows):
    """`bedtools closest -a A -b B -s -d -t all` on 6-column BED rows -> list of 13-column rows."""
    out = []
    ncol_b = len(b_rows[0]) if b_rows else 6
    for a in a_rows:
        a_s, a_e = int(a[1]), int(a[2])
        best, hits = None, []
        for b in b_rows:
            if b[0] != a[0] or b[5] != a[5]:
                continue
            d, _ = _dist(a_s, a_e, int(b[1]), int(b[2]))
            if best is None or d < best:
                best, hits = d, [b]
            elif d == best:
                hits.append(b)
        if best is None:
            out.append(list(a) + _null_b(ncol_b) + ["-1"])
        else:
            for b in hits:
                out.append(list(a) + list(b) + [str(best)])
    return out


def closest_signed_k(a_rows, b_rows, k=5):
    """`bedtools closest -a A -b B -d -D ref -k K` -> rows A + B + signed distance."""
    out = []
    ncol_b = len(b_rows[0]) if b_rows else 6
    for a in a_rows:
        a_s, a_e = int(a[1]), int(a[2])
        cand = []
        for i, b in enumerate(b_rows):
            if b[0] != a[0]:
                continue
            d, side = _dist(a_s, a_e, int(b[1]), int(b[2]))
            cand.append((d, i, d * (side if side else 1), b))
        if not cand:
            out.append(list(a) + _null_b(ncol_b) + ["-1"])
            continue
        cand.sort(key=lambda c: (c[0], c[1]))
        cut = cand[min(k, len(cand)) - 1][0]
        for d, _, sd, b in cand:
            if d <= cut:
                out.append(list(a) + list(b) + [str(sd)])
    return out


def merge_collapse(rows, d=0, col=3, delim=","):
    """`bedtools merge -d D -c (col+1) -o collapse` on sorted rows -> [chrom, start, end, collapsed]."""
    out = []
    for r in rows:
        s, e = int(r[1]), int(r[2])
        if out and out[-1][0] == r[0] and s <= out[-1][2] + d:
            out[-1][2] = max(out[-1][2], e)
            out[-1][3].append(r[col])
        else:
            out.append([r[0], s, e, [r[col]]])
    return [[c, str(s), str(e), delim.join(v)] for c, s, e, v in out]


def getfasta(seq_lookup, chrom, start, end):
    """`bedtools getfasta -fi FASTA -bed <chrom start end>` -> (header, sequence)."""
    s = seq_lookup(chrom)
    return "%s:%d-%d" % (chrom, start, end), s[start:end]


def intersect_wao(a_rows, b_rows):
    """`bedtools intersect -a A -b B -wao`: one row per (A, overlapping B) with the overlap in bp; an A
    feature without overlap gets `.`/-1 filler columns and 0.  (B is grouped by chromosome once: the annotation
    step intersects thousands of library hits with one ALT hit per contig.)"""
    out = []
    ncol_b = len(b_rows[0]) if b_rows else 6
    by_chrom = {}
    for b in b_rows:
        by_chrom.setdefault(b[0], []).append((int(b[1]), int(b[2]), b))
    for a in a_rows:
        a_s, a_e = int(a[1]), int(a[2])
        hit = False
        for b_s, b_e, b in by_chrom.get(a[0], ()):
            ov = min(a_e, b_e) - max(a_s, b_s)
            if ov > 0:
                out.append(list(a) + list(b) + [str(ov)])
                hit = True
        if not hit:
            out.append(list(a) + _null_b(ncol_b) + ["0"])
    return out


def merge_distinct(rows, d, cols, delim="|"):
    """`bedtools merge -d D -c cols -o distinct,... -delim DELIM` on sorted rows (cols are 0-based);
    `distinct` lists the unique values in lexicographic order."""
    groups = []
    for r in rows:
        s, e = int(r[1]), int(r[2])
        if groups and groups[-1][0] == r[0] and s <= groups[-1][2] + d:
            groups[-1][2] = max(groups[-1][2], e)
            groups[-1][3].append(r)
        else:
            groups.append([r[0], s, e, [r]])
    return [[c, str(s), str(e)] + [delim.join(sorted(set(r[k] for r in rs))) for k in cols] for c, s, e, rs in groups]
