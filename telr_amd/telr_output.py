"""Stage-4 output writers: final report (JSON), TE / contig FASTA, VCF and BED.

Mirror of the reference's output module (src/telr/TELR_output.py): `generate_output` :10-297,
`write_bed` :300-310, `write_vcf` :313-426, `get_contig_info` :429-438.  Same function names,
argument order and files written, so the golden test (tests/test_golden_output.py) diffs the files
byte for byte against what the reference wrote for the same inputs (tools/capture_goldens.py G4).

Behaviour kept on purpose, because downstream users parse these files:
  * the VCF sample column is `genotype:num_sv_reads:num_ref_reads` under the FORMAT string
    `GT:DR:DV` (the DR/DV order is swapped in the reference, TELR_output.py:321);
  * the reference builds the VCF body through a pandas DataFrame, so a column that mixes numbers
    with missing values is rendered as floats (`TSD_LEN=5.0`, `AF=nan`), a text column with a
    missing value renders `None` inside INFO, and only whole missing cells become `NA`
    (:313-372).  `_column_text` restates those rendering rules; pandas is not imported;
  * the VCF ID column is the row number; POS is `start + 1`; REF is `N`;
  * the TE sequence is reverse-complemented when the contig annotation says the TE is on `-`
    (:160-165); TSD sequences are upper-cased (:154-155);
  * `support` is `both_sides` only when both flank alignments were kept (:263-266).
"""
import datetime
import json
import os

from .fasta import revcomp

REPORT_KEYS = ("type", "ID", "chrom", "start", "end", "family", "strand", "support", "tsd_length", "tsd_sequence",
               "te_sequence", "genotype", "num_sv_reads", "num_ref_reads", "allele_frequency")
EXPANDED_KEYS = REPORT_KEYS + ("gap_between_flank", "te_length", "contig_id", "contig_length", "contig_te_start", "contig_te_end") + tuple(
    "%s_flank_%s" % (side, f) for side in ("5p", "3p")
    for f in ("align_coord", "mapping_quality", "num_residue_matches", "alignment_block_length", "sequence_identity"))
COV_KEYS = ("te_5p_cov", "te_3p_cov", "flank_5p_cov", "flank_3p_cov", "te_5p_cov_rc", "te_3p_cov_rc", "flank_5p_cov_rc", "flank_3p_cov_rc")

VCF_META = (
    '##INFO=<ID=END,Number=1,Type=Integer,Description="End position of the structure variant">',
    '##INFO=<ID=SVTYPE,Number=1,Type=String,Description="Type of structure variant">',
    '##INFO=<ID=STRANDS,Number=A,Type=String,Description="Strand orientation">',
    '##INFO=<ID=AF,Number=A,Type=Float,Description="Allele Frequency">',
    '##INFO=<ID=FAMILY,Number=1,Type=String,Description="TE family">',
    '##INFO=<ID=RE,Number=1,Type=Integer,Description="read support">',
    '##INFO=<ID=SUPPORT_TYPE,Number=1,Type=String,Description="single_side or both_sides">',
    '##INFO=<ID=TSD_LEN,Number=1,Type=String,Description="Length of the TSD sequence if available">',
    '##INFO=<ID=TSD_SEQ,Number=1,Type=String,Description="TSD sequence if available">',
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
    '##FORMAT=<ID=DR,Number=1,Type=Integer,Description="# high-quality reference reads">',
    '##FORMAT=<ID=DV,Number=1,Type=Integer,Description="# high-quality variant reads">',
)
VCF_COLUMNS = ("CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT", "SAMPLE")


# ---- loaders (the reference reads these from files between stages) ----

def _fasta_records(path):
    """(full header line without '>', sequence) per record, in file order."""
    hdr, buf = None, []
    with open(path, "r") as fh:
        for line in fh:
            line = line.rstrip("\n")
            if line.startswith(">"):
                if hdr is not None:
                    yield hdr, "".join(buf)
                hdr, buf = line[1:], []
            elif line:
                buf.append(line.strip())
    if hdr is not None:
        yield hdr, "".join(buf)


def _rec_id(header):
    f = header.split()
    return f[0] if f else ""


def load_contig_lengths(contig_fa):
    """contig name -> length from the ` len=N` description field of the merged contig FASTA (:23-31)."""
    out = {}
    for hdr, _ in _fasta_records(contig_fa):
        out[_rec_id(hdr)] = int(hdr.split(" ")[1].replace("len=", ""))
    return out


def load_te_strands(contig_te_annotation):
    """contig name -> '+', '-' or '.' from column 6 of the contig TE annotation BED (:33-43)."""
    out = {}
    with open(contig_te_annotation, "r") as fh:
        for line in fh:
            f = line.replace("\n", "").split("\t")
            out[f[0]] = f[5] if f[5] in ("+", "-") else "."
    return out


def load_sniffles_info(vcf_parsed):
    """contig name -> genotype / read counts from the parsed SV table; blanks inside fields are dropped (:45-60)."""
    out = {}
    with open(vcf_parsed, "r") as fh:
        for line in fh:
            f = line.replace("\n", "").replace(" ", "").split("\t")
            out["_".join(f[0:3])] = {"gt": f[10], "alt_count": f[12], "ref_count": f[11]}
    return out


def load_te_seqs(te_fa):
    return {_rec_id(h): s for h, s in _fasta_records(te_fa)}


# ---- report assembly ----

def build_reports(liftover_report, te_freq_dict, te_seqs, sniffles_info, contig_te_strand, contig_length):
    """-> (final_report, final_report_expanded, contig_ids) for the non-reference rows of the liftover report (:73-276)."""
    final, expanded, contig_ids = [], [], set()
    for item in liftover_report:
        rep = item["report"]
        if rep["type"] != "non-reference":
            continue
        ins_name = item["genome1_coord"]
        contig_id, te_coord = ins_name.split(":")[0], ins_name.split(":")[1]
        contig_ids.add(contig_id)
        row = dict.fromkeys(REPORT_KEYS)
        for k in ("type", "chrom", "start", "end", "family", "strand"):
            row[k] = rep[k]
        row["ID"] = "_".join([rep["chrom"], str(rep["start"]), str(rep["end"]), rep["family"]])
        row["tsd_length"] = rep["TSD_length"]
        if rep["TSD_sequence"]:
            row["tsd_sequence"] = rep["TSD_sequence"].upper()
        te_seq = str(te_seqs[ins_name])
        row["te_sequence"] = revcomp(te_seq) if contig_te_strand[contig_id] == "-" else te_seq
        sv = sniffles_info[contig_id]
        row["genotype"], row["num_sv_reads"], row["num_ref_reads"] = sv["gt"], sv["alt_count"], sv["ref_count"]
        freq = te_freq_dict[contig_id]
        row["allele_frequency"] = freq["freq"]

        ex = dict.fromkeys(EXPANDED_KEYS)
        for k in COV_KEYS:
            ex[k] = freq[k]
        ex["contig_length"] = contig_length[contig_id]
        ex["gap_between_flank"] = rep["gap"]
        ex["contig_id"] = contig_id
        ex["te_length"] = len(row["te_sequence"])
        ex["contig_te_start"], ex["contig_te_end"] = (int(x) for x in te_coord.split("-")[:2])
        for k in EXPANDED_KEYS[-10:]:
            ex[k] = rep[k]
        both = ex["5p_flank_align_coord"] is not None and ex["3p_flank_align_coord"] is not None
        row["support"] = "both_sides" if both else "single_side"
        ex.update(row)
        final.append(row)
        expanded.append(ex)
    return final, expanded, contig_ids


# ---- writers ----

def write_bed(final_report, bed):
    with open(bed, "w") as out:
        for r in final_report:
            out.write("\t".join([r["chrom"], str(r["start"]), str(r["end"]), r["family"], ".", r["strand"]]) + "\n")


def _is_num(v):
    return isinstance(v, (int, float)) and not isinstance(v, bool)


def _column_text(values):
    """How a DataFrame column built from these Python values prints cell by cell inside a row-wise
    `str()`: all-int stays int; numbers mixed with floats or with missing values become floats
    (missing -> 'nan'); anything else is the Python `str` of the object (missing -> 'None')."""
    present = [v for v in values if v is not None]
    if present and all(_is_num(v) for v in present):
        if len(present) == len(values) and all(isinstance(v, int) for v in present):
            return [str(v) for v in values]
        return ["nan" if v is None else repr(float(v)) for v in values]
    return [str(v) for v in values]


def _csv_field(s):
    """csv QUOTE_MINIMAL with a tab delimiter, as `DataFrame.to_csv(sep='\\t')` applies it."""
    if any(c in s for c in '\t"\r\n'):
        return '"' + s.replace('"', '""') + '"'
    return s


def vcf_body(final_report):
    """VCF data lines (no trailing newline each) for the final report rows (:313-372, :425-426)."""
    if not final_report:
        return []
    col = {k: _column_text([r.get(k) for r in final_report])
           for k in ("end", "family", "strand", "support", "num_sv_reads", "allele_frequency", "tsd_length", "tsd_sequence")}
    lines = []
    for i, r in enumerate(final_report):
        info = ("SVTYPE=INS;END=%s;FAMILY=%s;STRANDS=%s;SUPPORT_TYPE=%s;RE=%s;AF=%s;TSD_LEN=%s;TSD_SEQ=%s"
                % tuple(col[k][i] for k in ("end", "family", "strand", "support", "num_sv_reads", "allele_frequency", "tsd_length", "tsd_sequence")))
        parts = (r.get("genotype"), r.get("num_sv_reads"), r.get("num_ref_reads"))
        sample = "NA" if any(p is None for p in parts) else ":".join(parts)
        alt = "NA" if r.get("te_sequence") is None else r["te_sequence"]
        chrom = "NA" if r.get("chrom") is None else r["chrom"]
        fields = [chrom, str(r["start"] + 1), str(i), "N", alt, ".", "PASS", info, "GT:DR:DV", sample]
        lines.append("\t".join(_csv_field(f) for f in fields))
    return lines


def write_vcf(input, ref, ref_info, out_vcf, today=None):
    """`today` (a date or string) overrides the `##fileDate` stamp; default is the current date as in the reference."""
    stamp = datetime.date.today() if today is None else today
    with open(out_vcf, "w") as vcf:
        vcf.write("##fileformat=VCFv4.1\n##fileDate=%s\n##source=TELR\n##reference=%s\n" % (stamp, ref))
        vcf.write("\n".join(ref_info) + "\n")
        for line in VCF_META:
            vcf.write(line + "\n")
        vcf.write("#" + "\t".join(VCF_COLUMNS) + "\n")
        for line in vcf_body(input):
            vcf.write(line + "\n")


def write_fai(reference):
    """`samtools faidx` (called at :431): name, length, byte offset of the first base, bases per
    line, bytes per line.  Raises on ragged line lengths like samtools does."""
    rows = []
    with open(reference, "rb") as fh:
        pos, cur = 0, None
        for raw in fh:
            if raw.startswith(b">"):
                if cur:
                    rows.append(cur)
                cur = [raw[1:].split()[0].decode() if raw[1:].split() else "", 0, pos + len(raw), 0, 0, False]
            elif cur is not None:
                n = len(raw.rstrip(b"\r\n"))
                if n and cur[5]:
                    raise ValueError("different line length in sequence '%s'" % cur[0])
                if n == 0:
                    cur[5] = True
                elif cur[3] == 0:
                    cur[3], cur[4] = n, len(raw)
                elif n > cur[3]:
                    raise ValueError("different line length in sequence '%s'" % cur[0])
                elif n < cur[3]:
                    cur[5] = True
                cur[1] += n
            pos += len(raw)
        if cur:
            rows.append(cur)
    with open(reference + ".fai", "w") as out:
        for name, length, off, lb, lw, _ in rows:
            out.write("%s\t%d\t%d\t%d\t%d\n" % (name, length, off, lb, lw))


def get_contig_info(reference):
    """`##contig` header lines from the reference's .fai (:429-438); the index is written here when absent."""
    if not os.path.isfile(reference + ".fai"):
        write_fai(reference)
    info = []
    with open(reference + ".fai", "r") as fh:
        for line in fh:
            f = line.replace("\n", "").split("\t")
            info.append("##contig=<ID={},length={}>".format(f[0], f[1]))
    return info


def _write_fasta_record(out, header, seq, width=60):
    out.write(">" + header + "\n")
    for i in range(0, len(seq), width):
        out.write(seq[i:i + width] + "\n")


def generate_output(liftover_report_path, te_freq_dict, te_fa, vcf_parsed, contig_te_annotation, contig_fa, out, sample_name, ref, today=None):
    """Same arguments as the reference (:10-20).  `liftover_report_path` may also be the report list itself
    (the in-memory hand-over from `telr_liftover.liftover`).  Returns (final_report, final_report_expanded)."""
    if isinstance(liftover_report_path, (list, tuple)):
        liftover_report = list(liftover_report_path)
    else:
        with open(liftover_report_path) as fh:
            liftover_report = json.load(fh)
    final, expanded, contig_ids = build_reports(
        liftover_report, te_freq_dict, load_te_seqs(te_fa), load_sniffles_info(vcf_parsed),
        load_te_strands(contig_te_annotation), load_contig_lengths(contig_fa))

    stem = os.path.join(out, sample_name)
    with open(stem + ".telr.json", "w") as fh:
        json.dump(final, fh, indent=4, sort_keys=False)
    with open(stem + ".telr.expanded.json", "w") as fh:
        json.dump(expanded, fh, indent=4, sort_keys=False)
    with open(stem + ".telr.te.fasta", "w") as fh:
        for r in final:
            fh.write(">%s_%s_%s#%s\n%s\n" % (r["chrom"], r["start"], r["end"], r["family"], r["te_sequence"]))
    with open(stem + ".telr.contig.fasta", "w") as fh:
        for hdr, seq in _fasta_records(contig_fa):
            if _rec_id(hdr) in contig_ids:
                _write_fasta_record(fh, hdr, seq)
    write_vcf(final, ref, get_contig_info(ref), stem + ".telr.vcf", today=today)
    write_bed(final, stem + ".telr.bed")
    return final, expanded
