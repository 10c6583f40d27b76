"""Candidate-locus table (`<sample>.vcf_filtered.tsv`) handling: schema, merge of nearby calls, helpers.

Mirror of the table-side half of the reference's SV module: `merge_vcf` src/telr/TELR_sv.py:84-140, `string2int`
:143-150, `average` :153-156, `write_ins_seqs` :328-334, `id_merge` :337-341, `get_unique_list` :343-348, `af_sum`
:351-355, and `create_loci_set` src/telr/TELR_utility.py:44-50.  Calling Sniffles and RepeatMasker (the other half of
that module) is outside the alignment path and not provided.

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
