"""Candidate-locus table (`<sample>.vcf_filtered.tsv`) handling: schema, merge of nearby calls, helpers.

Mirror of the table-side half of the reference's SV module: `merge_vcf` src/telr/TELR_sv.py:84-140, `string2int`
:143-150, `average` :153-156, `swap_coordinate` :183-190, `rm_vcf_redundancy` :193-228, `filter_vcf` :231-324,
`write_ins_seqs` :328-334, `id_merge` :337-341, `get_unique_list` :343-348, `af_sum` :351-355, and `create_loci_set`
src/telr/TELR_utility.py:44-50.  Calling Sniffles / bcftools (`detect_sv`, the query in `parse_vcf`) is outside the
alignment path and not provided.  `filter_vcf` keeps the reference's table arithmetic and takes the TE intervals of the
ALT sequences from a `screen`: `gff_screen` reads a RepeatMasker `.out.gff` (the reference's source, :256-295),
`engine_screen` (opt-in: it changes which loci pass) maps the TE library onto the ALT sequences
with the HIP engine instead.

The 14 columns are addressed by index by every later stage; COLUMNS names them.

Kept on purpose:
  * the representative call of a merged group is chosen with `max()` over the SV lengths *as text*
    (lexicographic: "999" beats "1000"), TELR_sv.py:103-104;
  * merged start/end are Python-`round`ed means (ties to even), coverage is a float sum, AF a float sum capped to the
    integer 1, alt_count the number of distinct supporting reads (:100-115).
One deliberate difference: distinct read names keep first-appearance order instead of `set` iteration order, which in
the reference depends on PYTHONHASHSEED; the set of names, and therefore alt_count, is identical.
"""
COLUMNS = ("chrom", "start", "end", "sv_length", "coverage", "sniffles_af", "sv_id", "ins_seq", "reads", "filter",
           "genotype", "ref_count", "alt_count", "ins_te_prop")


def string2int(lst, integer=True):
    """in-place conversion of a list of numerals, returned for chaining"""
    conv = int if integer else float
    for i, v in enumerate(lst):
        lst[i] = conv(v)
    return lst


def average(lst):
    nums = string2int(lst.split(";"))
    return round(sum(nums) / len(nums))


def af_sum(nums):
    total = sum(nums)
    return 1 if total > 1 else total


def get_unique_list(list1):
    return list(dict.fromkeys(list1))


def id_merge(strings):
    return ",".join(get_unique_list(",".join(strings).split(",")))


def read_locus_table(path):
    """-> list of 14+-column rows (lists of str) in file order"""
    with open(path, "r") as fh:
        return [line.replace("\n", "").split("\t") for line in fh if line.strip()]


def locus_name(row):
    return "_".join(row[0:3])


def create_loci_set(vcf_parsed):
    return set(locus_name(r) for r in read_locus_table(vcf_parsed))


def bedtools_merge_rows(rows, window=20):
    """`bedtools merge -o collapse -c 2,...,14 -delim ";" -d WINDOW` on the (already sorted) table:
    chrom, merged start, merged end, then the 13 collapsed columns."""
    out = []
    for chrom, s, e, members in _groups(rows, window):
        out.append([chrom, str(s), str(e)] + [";".join(m[k] for m in members) for k in range(1, 14)])
    return out


def _groups(rows, window):
    cur = None
    for r in rows:
        s, e = int(r[1]), int(r[2])
        if cur is not None and cur[0] == r[0] and s <= cur[2] + window:
            cur[2] = max(cur[2], e)
            cur[3].append(r)
        else:
            if cur is not None:
                yield cur
            cur = [r[0], s, e, [r]]
    if cur is not None:
        yield cur


def collapse_group(entry):
    """one line of the merge intermediate -> one 14-column locus row (TELR_sv.py:97-138)"""
    if ";" not in entry[3]:
        return [entry[0]] + entry[3:]
    lens = entry[5].split(";")
    idx = lens.index(max(lens))
    reads = ",".join(get_unique_list(entry[10].replace(";", ",").split(",")))
    pick = lambda k: entry[k].split(";")[idx]
    return [entry[0], str(average(entry[3])), str(average(entry[4])), str(lens[idx]),
            str(sum(string2int(entry[6].split(";"), integer=False))), str(af_sum(string2int(entry[7].split(";"), integer=False))),
            pick(8), pick(9), reads, pick(11), pick(12), str(pick(13)), str(len(reads.split(","))), str(pick(15))]


def merge_rows(rows, window=20):
    return [collapse_group(e) for e in bedtools_merge_rows(rows, window)]


def merge_vcf(vcf_in, vcf_out, window=20):
    rows = merge_rows(read_locus_table(vcf_in), window)
    with open(vcf_out, "w") as out:
        for r in rows:
            out.write("\t".join(r) + "\n")


def write_ins_seqs(vcf, out):
    with open(out, "w") as fh:
        for r in read_locus_table(vcf):
            fh.write(">" + locus_name(r) + "\n" + r[7] + "\n")


def swap_coordinate(vcf_in, vcf_out):
    """start/end exchanged where end < start (compared as integers, written back as the original text)"""
    with open(vcf_out, "w") as out:
        for r in read_locus_table_raw(vcf_in):
            if int(r[2]) < int(r[1]):
                r[1], r[2] = r[2], r[1]
            out.write("\t".join(r) + "\n")


def read_locus_table_raw(path):
    """every line, blank ones included, split on tabs (swap_coordinate does not skip anything)"""
    with open(path, "r") as fh:
        return [line.replace("\n", "").split("\t") for line in fh]


def _typed_column(values):
    """column type inference as the table reader of the reference does it (pandas.read_csv on a headerless TSV):
    all integers -> int, else all numbers -> float, else text as is.  Missing-value handling is not reproduced
    (the parsed VCF table has no empty fields)."""
    for conv in (int, float):
        try:
            return [conv(v) for v in values]
        except ValueError:
            pass
    return list(values)


def _cell_text(v):
    return repr(v) if isinstance(v, float) else str(v)


def _column_sum(vals):
    return sum(vals) if not isinstance(vals[0], str) else "".join(vals)


def dedup_rows(rows):
    """rows sharing (chrom, start, end) collapse to one: first length / id / sequence / filter / genotype, summed
    coverage and read counts, capped AF sum, union of read names (TELR_sv.py:209-227).  Returns typed rows sorted
    by (chrom, start, end) as the group-by does; an AF column whose every group sum was capped prints as the integer."""
    cols = [_typed_column([r[k] for r in rows]) for k in range(13)]
    groups = {}
    for i in range(len(rows)):
        groups.setdefault((cols[0][i], cols[1][i], cols[2][i]), []).append(i)
    out = []
    for key in sorted(groups):
        m = groups[key]
        first = lambda k: cols[k][m[0]]
        out.append([key[0], key[1], key[2], first(3), _column_sum([cols[4][i] for i in m]),
                    af_sum([cols[5][i] for i in m]), first(6), first(7), id_merge([str(cols[8][i]) for i in m]),
                    first(9), first(10), _column_sum([cols[11][i] for i in m]), _column_sum([cols[12][i] for i in m])])
    if any(isinstance(r[5], float) for r in out):
        for r in out:
            r[5] = float(r[5])
    return out


def rm_vcf_redundancy(vcf_in, vcf_out):
    rows = dedup_rows(read_locus_table(vcf_in))
    with open(vcf_out, "w") as out:
        for r in rows:
            out.write("\t".join(_cell_text(v) for v in r) + "\n")


def gff_screen(gff_path):
    """RepeatMasker `.out.gff` -> merged TE intervals per ALT sequence: `bedtools sort` then `bedtools merge` on a GFF
    (1-based inclusive features, 0-based starts printed, overlapping and book-ended features joined), TELR_sv.py:283-295"""
    feats = []
    with open(gff_path, "r") as fh:
        for line in fh:
            if line.startswith("#") or not line.strip():
                continue
            f = line.rstrip("\n").split("\t")
            feats.append((f[0], int(f[3]) - 1, int(f[4])))
    return merge_intervals(feats)


def merge_intervals(feats):
    out = []
    for name, s, e in sorted(feats, key=lambda f: (f[0], f[1])):
        if out and out[-1][0] == name and s <= out[-1][2]:
            out[-1][2] = max(out[-1][2], e)
        else:
            out.append([name, s, e])
    return out


def engine_screen(backend, presets="ont", min_len=0):
    """-> screen(ins_fasta, te_library, thread) that maps every TE consensus onto every ALT sequence in one engine call
    (library sketched once, chains ranked per ALT sequence) and returns the merged target intervals.  Replaces the
    RepeatMasker run of TELR_sv.py:253-281; hit sets differ from RepeatMasker's, hence opt-in."""
    from . import fasta
    from ._abi import MF_PER_TARGET
    from .presets import preset

    def screen(ins_fasta, te_library, thread):
        names, seqs = fasta.read_fasta(ins_fasta)
        _, lib = fasta.read_fasta(te_library)
        if not names or not lib:
            return []
        io, mo = preset("map-ont" if presets == "ont" else "map-pb")
        mo = mo.copy(); mo.flags |= MF_PER_TARGET
        res = backend.index(list(seqs), io).map(list(lib), mo)
        return merge_intervals([(names[a["tid"]], int(a["ts"]), int(a["te"])) for a in res.alns
                                if int(a["te"]) - int(a["ts"]) >= min_len])
    return screen


def te_proportions(merged, contig_len):
    """per ALT sequence: the SUM over its merged TE intervals of round(interval length / sequence length, 2)
    (rounded per interval, then added as floats, TELR_sv.py:298-309)"""
    props = {}
    for name, s, e in merged:
        p = round((int(e) - int(s)) / contig_len[name], 2)
        props[name] = props[name] + p if name in props else p
    return props


def filter_vcf(ins, ins_filtered, te_library, out, sample_name, thread, loci_eval, screen=None):
    """keep the loci whose ALT sequence carries TE sequence and append the TE proportion as column 14; the others are
    appended to `loci_eval` as "VCF sequence not repeatmasked".  `screen(ins_fasta, te_library, thread)` supplies the
    merged TE intervals; there is no default (RepeatMasker is a hand-off point): pass `engine_screen(engine)` or
    `lambda *a: gff_screen(path)`."""
    import os
    if screen is None:
        raise ValueError("filter_vcf needs a screen: engine_screen(backend) or a RepeatMasker GFF via gff_screen")
    ins_seqs = os.path.join(out, sample_name.replace("+", "plus") + ".vcf_ins.fasta")
    write_ins_seqs(ins, ins_seqs)
    rows = read_locus_table(ins)
    contig_len = {locus_name(r): len(r[7]) for r in rows}
    props = te_proportions(screen(ins_seqs, te_library, thread), contig_len)
    with open(ins_filtered, "w") as fh:
        for r in rows:
            if locus_name(r) in props:
                fh.write("\t".join(r) + "\t" + str(props[locus_name(r)]) + "\n")
    with open(loci_eval, "a") as fh:
        seen = set()
        for r in rows:
            n = locus_name(r)
            if n not in props and n not in seen:
                seen.add(n)
                fh.write(n + "\tVCF sequence not repeatmasked\n")
