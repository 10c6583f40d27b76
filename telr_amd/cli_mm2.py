"""`telr-mm2`: command-line shim that accepts the seven aligner argv shapes of the reference and answers
them with the HIP engine, writing SAM / PAF to stdout exactly where the reference redirects stdout.

  S1  ngmlr -r R -q Q -x {ont,pacbio} -t T --rg-id S --rg-sm S --rg-lb {ont,pb} --no-progress   (TELR_alignment.py:31-51)
  S2  minimap2 --cs --MD -Y -L -ax {map-ont,map-pb} R Q                                       (TELR_alignment.py:69-82)
  S3  minimap2 -t N -ax P -r2k CNS READS                                                       (TELR_assembly.py:199-212)
  S4  minimap2 -cx P --secondary=no -v 0 SUBJ QRY                                              (TELR_te.py:68-78)
  S5  minimap2 -cx P CONTIG LIB -v 0 -t T                                                      (TELR_te.py:119-132)
  S6  minimap2 -a -x P -v 0 SUBJ QRY                                                           (TELR_te.py:504-506)
  S7  minimap2 -cx asm10 -v 0 -N 10 REF FLANK                                                  (TELR_liftover.py:253-266)

usage:  python -m telr_amd.cli_mm2 minimap2 <args...>   |   python -m telr_amd.cli_mm2 ngmlr <args...>
Exit code 0 and empty stdout when nothing maps (the reference treats an empty PAF as "locus not passed").
"""
import sys


def parse_argv(argv):
    """-> dict(tool, preset, sam, cigar, md, cs, softclip, secondary, best_n, bw, target, query, rg, threads)"""
    if not argv:
        raise SystemExit(__doc__)
    tool = argv[0].split("/")[-1]
    a = argv[1:]
    o = dict(tool=tool, preset=None, sam=False, cigar=False, md=False, cs=False, softclip=False, secondary=True, best_n=None,
             bw=None, target=None, query=None, rg=None, threads=None)
    if tool == "ngmlr":
        rg = {"id": None, "sm": None, "lb": None}
        i = 0
        while i < len(a):
            f = a[i]
            if f == "-r":
                o["target"] = a[i + 1]; i += 2
            elif f == "-q":
                o["query"] = a[i + 1]; i += 2
            elif f == "-x":
                if a[i + 1] not in ("ont", "pacbio"):
                    raise SystemExit("ngmlr -x must be ont or pacbio")
                o["preset"] = "ngmlr-" + a[i + 1]; i += 2
            elif f == "-t":
                o["threads"] = int(a[i + 1]); i += 2
            elif f in ("--rg-id", "--rg-sm", "--rg-lb"):
                rg[f[5:]] = a[i + 1]; i += 2
            elif f == "--no-progress":
                i += 1
            else:
                raise SystemExit("unsupported ngmlr option %r" % f)
        o["sam"] = o["cigar"] = o["md"] = True
        o["softclip"] = True
        if rg["id"]:
            o["rg"] = (rg["id"], rg["sm"] or rg["id"], rg["lb"] or "lib")
        if not o["preset"]:
            o["preset"] = "ngmlr-pacbio"
    elif tool == "minimap2":
        pos = []
        i = 0
        while i < len(a):
            f = a[i]
            if f == "--cs":
                o["cs"] = True; i += 1
            elif f == "--MD":
                o["md"] = True; i += 1
            elif f == "-Y":
                o["softclip"] = True; i += 1
            elif f == "-L" or f == "-a" or f == "-c":
                if f == "-a":
                    o["sam"] = o["cigar"] = True
                if f == "-c":
                    o["cigar"] = True
                i += 1
            elif f in ("-ax", "-cx", "-x"):
                if f == "-ax":
                    o["sam"] = o["cigar"] = True
                if f == "-cx":
                    o["cigar"] = True
                o["preset"] = a[i + 1]; i += 2
            elif f.startswith("--secondary="):
                o["secondary"] = f.split("=", 1)[1] not in ("no", "0", "false"); i += 1
            elif f == "-v":
                i += 2
            elif f == "-t":
                o["threads"] = int(a[i + 1]); i += 2
            elif f == "-N":
                o["best_n"] = int(a[i + 1]); i += 2
            elif f.startswith("-r") and len(f) > 2:
                v = f[2:].lower()
                o["bw"] = int(float(v[:-1]) * 1000) if v.endswith("k") else int(v); i += 1
            elif f == "-r":
                v = a[i + 1].lower()
                o["bw"] = int(float(v[:-1]) * 1000) if v.endswith("k") else int(v); i += 2
            elif f.startswith("-"):
                raise SystemExit("unsupported minimap2 option %r" % f)
            else:
                pos.append(f); i += 1
        if len(pos) != 2:
            raise SystemExit("expected <target.fa> <query.fa>")
        o["target"], o["query"] = pos
        if o["preset"] is None:
            o["preset"] = "map-ont"
    else:
        raise SystemExit("first argument must be minimap2 or ngmlr")
    return o


def run(argv, out_path="/dev/stdout", engine=None):
    """answer one argv shape (argv[0] = minimap2 | ngmlr) with the engine; SAM / PAF goes to `out_path`.  -> number of records"""
    o = parse_argv(argv)
    from .aligner import Engine
    from .presets import preset
    from .fasta import read_fasta
    io, mo = preset(o["preset"])
    if not o["secondary"]:
        mo.secondary = 0
    if o["best_n"] is not None:
        mo.best_n = o["best_n"]
    if o["bw"] is not None:
        mo.bw = o["bw"]
    if not o["cigar"]:
        mo.flags &= ~1
    tn, ts = read_fasta(o["target"])
    qn, qs = read_fasta(o["query"])
    eng = engine if engine is not None else Engine(0)
    ix = eng.index(ts, io)
    # S5 shape (library against ONE contig file) and every other shape are all-vs-all here; the batched
    # per-locus forms are in telr_amd.telr_te / telr_af / telr_liftover
    r = ix.map_raw(qs, mo)
    try:
        n = int(eng.L.telr_result_count(r))
        if o["sam"]:
            ix.write_sam(r, qn, qs, tn, ts, out_path, md=o["md"], cs=o["cs"], softclip=o["softclip"], rg=o["rg"],
                         cmdline=" ".join([o["tool"]] + list(argv[1:])))
        else:
            ix.write_paf(r, qn, tn, out_path, with_cigar=o["cigar"])
    finally:
        ix.free_raw(r)
        ix.free()
    return n


def main(argv=None):
    sys.stdout.flush()
    run(sys.argv[1:] if argv is None else argv)
    return 0


if __name__ == "__main__":
    sys.exit(main())
