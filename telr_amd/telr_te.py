"""Contig TE annotation: the two aligner call sites of `annotate_contig` and the interval arithmetic
between them (reference src/telr/TELR_te.py:21-262).

  S4  Sniffles ALT sequence -> its contig   `minimap2 -cx P --secondary=no -v 0 contig.fa ins.fa` per locus
      (:48-85), PAF columns (0,7,8,5,11,4) -> seq2contig BED (:87-95);
  S5  TE library -> contig                  `minimap2 -cx P contig.fa LIB -v 0 -t T` per locus (:108-133),
      PAF columns (5,7,8,0,11,4) -> te2contig BED (:134-142);
  then `bedtools intersect -wao` (overlap > 10 bp), `sort`, `merge -d 10000 -c 4,6 -o distinct,distinct
  -delim "|"` (:143-236).
Here S4 is ONE engine call for all loci (query i sees only its contig) and S5 is ONE call in which the
library is sketched once and chains are ranked per contig (TELR_MF_PER_TARGET).  The RepeatMasker
re-annotation that follows in the reference (:267-370) is a hand-off point and is not reproduced.
"""
import numpy as np

from . import intervals as iv
from ._abi import MF_PER_TARGET
from .presets import preset


def _strand(a):
    return "-" if a["flags"] & 8 else "+"


def annotate_contig(backend, contig_names, contig_seqs, alt_seqs, lib_names, lib_seqs, presets="ont", contig_set=None):
    """-> (annotation BED rows [contig, start, end, families, ".", strand], seq2contig rows, te2contig rows)

    backend: telr_amd.aligner.Engine (or a test double with the same index()/map() surface).
    alt_seqs[i] is the Sniffles ALT sequence of locus i (None = locus skipped).
    contig_set: the contigs as a sequence set already on the device (same order as contig_seqs), indexed as it is."""
    io, mo = preset("map-ont" if presets == "ont" else "map-pb")
    ix = backend.index(contig_set if contig_set is not None else list(contig_seqs), io)
    # S4: --secondary=no, query i restricted to contig i
    mo4 = mo.copy(); mo4.secondary = 0
    q_idx = [i for i, s in enumerate(alt_seqs) if s]
    seq2contig = []
    passed = set()
    if q_idx:
        res = ix.map([alt_seqs[i] for i in q_idx], mo4, qtarget=np.array(q_idx, np.int32))
        for a in res.alns:
            li = q_idx[a["qid"]]
            seq2contig.append([contig_names[li], str(a["ts"]), str(a["te"]), contig_names[a["tid"]], str(a["mapq"]), _strand(a)])
            passed.add(li)
    # S5: the whole library against every passed contig, ranked per contig
    mo5 = mo.copy(); mo5.flags |= MF_PER_TARGET
    te2contig = []
    if passed:
        res = ix.map(list(lib_seqs), mo5)
        for a in res.alns:
            if a["tid"] in passed:
                te2contig.append([contig_names[a["tid"]], str(a["ts"]), str(a["te"]), lib_names[a["qid"]], str(a["mapq"]), _strand(a)])
        # the reference appends per-locus PAFs in locus order; inside a locus minimap2 prints per query
        order = {n: k for k, n in enumerate(contig_names)}
        te2contig.sort(key=lambda r: order[r[0]])
    # intersect -wao, keep overlaps > 10 bp, sort, merge -d 10000 distinct family / strand
    kept = [r[:6] for r in iv.intersect_wao(te2contig, seq2contig) if int(r[12]) > 10]
    merged = iv.merge_distinct(iv.bed_sort(kept), 10000, [3, 5], "|")
    ann = []
    for c, s, e, fam, strand in merged:
        ann.append([c, s, e, fam, ".", strand if strand in ("+", "-") else "."])
    return iv.bed_sort(ann), seq2contig, te2contig


# ---- RepeatMasker hand-off (reference src/telr/TELR_te.py:391-494, telr.py:132-144) ------------------------------------
# RepeatMasker itself is an external tool and stays one (hand-off H2).  What is restated here is the glue around it --
# `parse_rm_out` (RepeatMasker's GFF2 -> the GFF3 the pipeline reads), `gff3tobed` (-> the sorted TE BED of the reference
# genome that the liftover consults) -- and, new, a cache: the reference masks the WHOLE reference genome with the TE library
# on every run (telr.py:132-139; tens of minutes for dm6, hours for a human chromosome), although the result depends on
# nothing but the two files and RepeatMasker's flags.  `repeatmask(..., cache_dir=...)` keys the three output files by the
# SHA-256 of (reference bytes, library bytes, the flag string) and copies them back on a hit.
import hashlib
import os
import re
import shutil
import subprocess
import sys

RM_FLAGS = ["-gff", "-s", "-nolow", "-no_is", "-e", "ncbi"]          # TELR_te.py:397-404


# RepeatMasker's `-gff` output is GFF2: nine tab-separated columns, source `RepeatMasker`, and the hit's family inside the
# attribute column as  Target "Motif:<family>" <start in consensus> <end in consensus>.
_RM_TARGET = re.compile(r'Target\s+"?(?:Motif:)?([^"\s]+)"?')
NO_REPEATS = "There were no repetitive sequences detected"          # what RepeatMasker leaves in <ref>.out when nothing was masked


def _columns(path, want):
    """(line, columns) of every line of a GFF file that has at least `want` tab-separated columns"""
    with open(path) as fh:
        for line in fh:
            cols = line.rstrip("\n").split("\t")
            if len(cols) >= want:
                yield line, cols


def parse_rm_out(rm_gff, gff3):
    """RepeatMasker's .out.gff -> the GFF3 the pipeline reads: same coordinates, score, strand and phase, type
    `dispersed_repeat`, attribute `Target=<family>` (the job of TELR_te.py:472-494; same records, same bytes)."""
    out = []
    for line, c in _columns(rm_gff, 9):
        if "RepeatMasker" not in line:                               # header lines (##gff-version, ##date, ##sequence-region)
            continue
        m = _RM_TARGET.search(c[8])
        family = m.group(1) if m else c[8].split(" ")[1].replace('"Motif:', "").replace('"', "")
        out.append("%s\tRepeatMasker\tdispersed_repeat\t%s\tTarget=%s\n" % (c[0], "\t".join(c[3:8]), family))
    with open(gff3, "w") as fh:
        fh.writelines(out)


def gff3tobed(gff, bed):
    """the reference genome's TE annotation as BED6 -- 0-based start, family as the name, score `.` -- in `bedtools sort`
    order (the job of TELR_te.py:436-469).  A GFF3 whose first record carries no `Target=` attribute is refused."""
    rows = []
    for line, c in _columns(gff, 1):
        if "#" in line:
            continue
        attrs = dict(kv.split("=", 1) for kv in (c[8].split(";") if len(c) > 8 else []) if "=" in kv)
        if "Target" not in attrs:
            if not rows:
                print("Incorrect GFF3 format, please check README for expected format, exiting...")
                sys.exit(1)
            attrs["Target"] = rows[-1][3]                            # (an attribute-less later record keeps the previous family)
        rows.append([c[0], str(int(c[3]) - 1), c[4], attrs["Target"], ".", c[6]])
    with open(bed, "w") as fh:
        fh.write("".join("\t".join(r) + "\n" for r in iv.bed_sort(rows)))


def _sha256_file(path, h):
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 22), b""):
            h.update(blk)


def repeatmask_key(ref, library):
    h = hashlib.sha256()
    _sha256_file(ref, h); h.update(b"\0library\0"); _sha256_file(library, h); h.update(("\0" + " ".join(RM_FLAGS)).encode())
    return h.hexdigest()


def repeatmask(ref, library, outdir, thread, cache_dir=None, runner=subprocess.call):
    """TELR_te.py:391-433 with the same return value (masked FASTA path, GFF3 path or None) and the same failure behaviour
    (`sys.exit(1)`).  cache_dir: results of earlier runs on the same (reference, library) are copied instead of running
    RepeatMasker again; a fresh result is stored there.  runner: what executes the argv (tests pass a stand-in)."""
    os.makedirs(outdir, exist_ok=True)
    base = os.path.basename(ref)
    ref_rm, gff, gff3 = (os.path.join(outdir, base + x) for x in (".masked", ".out.gff", ".out.gff3"))
    key = repeatmask_key(ref, library) if cache_dir else None
    slot = os.path.join(cache_dir, key) if cache_dir else None
    if slot and os.path.isfile(os.path.join(slot, "done")):
        if os.path.isfile(os.path.join(slot, "masked")):
            shutil.copyfile(os.path.join(slot, "masked"), ref_rm); shutil.copyfile(os.path.join(slot, "gff3"), gff3)
            return ref_rm, gff3
        return ref, None                                             # cached: "no repetitive sequences detected"
    failed = False
    try:
        runner(["RepeatMasker", "-dir", outdir] + RM_FLAGS + ["-lib", library, "-pa", str(thread), ref])
        if os.path.exists(ref_rm):
            parse_rm_out(gff, gff3)
        else:
            # no masked FASTA: either nothing was found (RepeatMasker says so in <ref>.out) or the tool failed
            with open(os.path.join(outdir, base + ".out")) as fh:
                report = fh.read()
            if NO_REPEATS in report:
                print("No repetitive sequences detected")
                ref_rm, gff3 = ref, None
            else:
                failed = True
    except (OSError, IndexError, ValueError) as e:
        print(e)
        failed = True
    if failed:
        print("Repeatmasking failed, exiting...")
        sys.exit(1)
    if slot:
        tmp = slot + ".tmp%d" % os.getpid()
        os.makedirs(tmp, exist_ok=True)
        if gff3 is not None:
            shutil.copyfile(ref_rm, os.path.join(tmp, "masked")); shutil.copyfile(gff3, os.path.join(tmp, "gff3"))
        open(os.path.join(tmp, "done"), "w").close()
        try:
            os.replace(tmp, slot)
        except OSError:                                              # another run stored it meanwhile
            shutil.rmtree(tmp, ignore_errors=True)
    return ref_rm, gff3
