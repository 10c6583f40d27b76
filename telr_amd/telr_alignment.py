"""Stage-1 alignment, mirroring the reference's `src/telr/TELR_alignment.py`.

`alignment(bam, read, reference, out, sample_name, thread, method, presets)` keeps the reference's
signature and argument meaning (:9): `method` is "nglmr" (sic, the reference's spelling) or "minimap2",
`presets` is "ont" or "pacbio".  Instead of `ngmlr ... > tmp.sam` / `minimap2 --cs --MD -Y -L -ax ... >
sam` (:28-82) followed by `samtools sort` + `samtools index` (:103-114), the reads are mapped by the HIP
engine and the coordinate-sorted, indexed BAM is built on the device from what is resident there (reads,
reference, CIGARs: telr_write_bam_dev) -- the host parses the two input files (telr_fasta_load), packs and
uploads them, and moves the finished file image into `bam`, whose pages are being allocated while the reads
are still mapped (telr_bam_prepare).
"""
import atexit
import logging
import os
import sys
import threading
import time

from .aligner import Engine
from .fasta import read_fasta, load
from .presets import preset
from ._abi import MF_KEEP_CIGARS


def format_time(seconds):          # TELR_utility.py:34-41
    h, rem = divmod(seconds, 3600)
    m, s = divmod(rem, 60)
    return "%d:%02d:%02d" % (int(h), int(m), round(s))


_release_threads = []


def wait_release():
    """join the background releases of earlier alignment() calls (before Engine.close(), or at exit)"""
    while _release_threads:
        _release_threads.pop().join()


atexit.register(wait_release)


def alignment(bam, read, reference, out, sample_name, thread, method, presets, engine=None):
    logging.info("Start alignment...")
    start_time = time.time()
    if presets not in ("ont", "pacbio"):
        print("Read presets not recognized, please provide ont or pacbio, exiting...")
        sys.exit(1)
    if method == "nglmr":
        name = "ngmlr-ont" if presets == "ont" else "ngmlr-pacbio"
        rg = (sample_name, sample_name, "ont" if presets == "ont" else "pb")     # --rg-id/--rg-sm/--rg-lb (:32-49)
        cmd = "ngmlr -r %s -q %s -x %s -t %s" % (reference, read, presets, thread)
    elif method == "minimap2":
        name = "map-ont" if presets == "ont" else "map-pb"
        rg = None
        cmd = "minimap2 --cs --MD -Y -L -ax %s %s %s" % (name, reference, read)
    else:
        print("Alignment method not recognized, please provide ont or pacbio, exiting...")
        sys.exit(1)
    eng = engine or Engine(0)
    io, mo = preset(name)
    tm = {}                                       # seconds per phase of the last call: alignment.last_timings
    t0 = time.time()
    tf, qf = load(reference), load(read)          # None for gzip: the Python reader
    if tf is not None:
        tn, ts = tf.names_c, tf.triple
    else:
        tn, ts = read_fasta(reference)
    if qf is not None:
        qn, qs, n_bases = qf.names_c, qf.triple, int(qf.triple[2].sum())
    else:
        qn, qs = read_fasta(read)
        n_bases = sum(len(x) for x in qs)
    with_cs = method == "minimap2"
    tm["parse_files"] = time.time() - t0; t0 = time.time()
    ix = eng.index(ts, io)
    tm["pack_reference_build_index"] = time.time() - t0; t0 = time.time()
    qset = eng.seqset(qs)
    tm["pack_upload_reads"] = time.time() - t0; t0 = time.time()
    # about 0.85 bytes of BAM per read base with --cs --MD, 0.6 without cs, at level 1
    ix.bam_prepare(bam, int((0.95 if with_cs else 0.7) * n_bases) + (64 << 20))
    mo.flags |= MF_KEEP_CIGARS                   # the CIGAR array stays on the device as well: the BAM writer reads it there
    r = None
    try:
        r = ix.map_raw(qset, mo)
        tm["map"] = time.time() - t0; t0 = time.time()
        ix.write_bam_device(r, qset, qn, tn, bam, md=True, cs=with_cs, softclip=True, rg=rg, cmdline=cmd, index=True, level=1)
        tm["sorted_bam"] = time.time() - t0; t0 = time.time()
    except BaseException:
        # nothing half-made at the output path: the pre-sized file of bam_prepare goes with its sink (the writer itself
        # unlinks what it could not finish), so that the existence test below -- the reference's -- means what it says
        ix.bam_discard()
        for leftover in (bam, bam + ".bai"):
            if os.path.exists(leftover):
                os.unlink(leftover)
        raise
    finally:
        if r is not None:
            ix.free_raw(r)
        # giving the rest back -- hipFree of the packed reads and of the index (0.12 s for a 30x set), the parsed files or their
        # mappings -- is not something the caller has to wait for with its BAM finished: sequence sets and indexes are plain device
        # memory without context state.  The thread is joined at interpreter exit and by wait_release() (a caller that closes
        # the engine right away).

        def _drop(files, reads, index):
            reads.free()
            index.free()
            for f in files:
                if f is not None:
                    f.close()
        th = threading.Thread(target=_drop, args=((tf, qf), qset, ix), daemon=False)
        _release_threads.append(th)
        th.start()
    tm["release"] = time.time() - t0
    # (the engine keeps its grow-only mapping scratch, 150-200 GB for a 30x read set: a caller that goes on to the per-locus
    # stages with the same engine gives it back with Engine.release_scratch() -- 0.2 s -- when it knows stage 1 will not run again)
    alignment.last_timings = tm
    if os.path.isfile(bam) is False:
        sys.stderr.write("Sorted and indexed BAM file does not exist, exiting...\n")
        sys.exit(1)
    logging.info("First alignment finished in " + format_time(time.time() - start_time))
