"""Sharding of the path across the GPUs of one node.

* stage 1: reads are dealt to ranks by cumulative bases (length-sorted, snake order) against a
  replicated index — no collective on the data path;
* stages 2-4: candidate loci are assigned by LPT (longest processing time first) on their contig
  + read bases; every rank produces fixed-width result rows for its loci and ONE all-gather of
  fixed-capacity blocks merges them (RCCL over xGMI on GPUs, `backend="nccl"`; gloo in the CPU tests).
  Variable-length payloads (sequences, CIGARs) stay on the owning rank;
* between the two (only when the reads themselves are sharded): the window reads of a locus sit on whichever
  ranks mapped them and travel to the locus' owner in one all-to-all (`exchange_window_reads_packed`).
The reference's only parallelism on this path is `multiprocessing.Pool(processes=thread)` over loci
(src/telr/TELR_assembly.py:70-71, TELR_te.py:644-646, TELR_liftover.py:1049-1052).
"""
import numpy as np

TYPE_CODES = {"unlifted": 0, "non-reference": 1, "reference": 2}
TYPE_NAMES = {v: k for k, v in TYPE_CODES.items()}

LOCUS_ROW = np.dtype([
    ("locus_id", np.int32), ("status", np.int32), ("chrom_id", np.int32), ("start", np.int32), ("end", np.int32),
    ("strand", np.int8), ("type", np.int8), ("n_family", np.int8), ("pad", np.int8),
    ("gap", np.int32), ("tsd_len", np.int32), ("support", np.int32), ("family_id", np.int32, (4,)),
    ("n_sv_reads", np.int32), ("n_ref_reads", np.int32), ("medians", np.float32, (8,)), ("af", np.float64),
], align=True)
NONE_I32 = np.int32(-2147483648)          # "None" for integer columns
assert LOCUS_ROW.itemsize == 104, LOCUS_ROW.itemsize


def shard_reads(lengths, world):
    """-> list (per rank) of read indices: length-sorted, dealt in snake order so every rank gets
    the same number of reads (+-1) and nearly the same number of bases."""
    order = np.argsort(-np.asarray(lengths, dtype=np.int64), kind="stable")
    out = [[] for _ in range(world)]
    for k, i in enumerate(order):
        r = k % (2 * world)
        out[r if r < world else 2 * world - 1 - r].append(int(i))
    return [sorted(x) for x in out]


def shard_loci(costs, world):
    """LPT: loci by decreasing cost to the currently lightest rank -> list (per rank) of locus indices."""
    order = np.argsort(-np.asarray(costs, dtype=np.int64), kind="stable")
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        out[r].append(int(i)); load[r] += int(costs[i])
    return [sorted(x) for x in out]


def all_gather_rows(rows, dist=None, device=None, capacity=None):
    """rows: LOCUS_ROW array of this rank -> all ranks' rows sorted by locus_id.

    ONE collective: every rank contributes a fixed-capacity block [count:int64][capacity rows] to
    torch.distributed.all_gather_into_tensor.  `capacity` must be the same on all ranks: the largest shard of the
    (deterministic) locus assignment, which every rank computes for itself."""
    rows = np.ascontiguousarray(rows, dtype=LOCUS_ROW)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.sort(rows, order="locus_id")
    import torch
    world = dist.get_world_size()
    if capacity is None:
        raise ValueError("all_gather_rows at world size > 1 needs the common block capacity (the largest shard)")
    if len(rows) > capacity:
        raise ValueError("rank holds %d rows, block capacity is %d" % (len(rows), capacity))
    block = np.zeros(8 + capacity * LOCUS_ROW.itemsize, np.uint8)
    block[:8] = np.frombuffer(np.int64(len(rows)).tobytes(), np.uint8)
    block[8:8 + rows.nbytes] = rows.view(np.uint8).reshape(-1)
    send = torch.from_numpy(block).to(device) if device is not None else torch.from_numpy(block)
    recv = torch.empty(world * len(block), dtype=torch.uint8, device=send.device)
    dist.all_gather_into_tensor(recv, send)
    buf = recv.cpu().numpy()
    parts = []
    for r in range(world):
        b = buf[r * len(block):(r + 1) * len(block)]
        k = int(np.frombuffer(b[:8].tobytes(), np.int64)[0])
        parts.append(np.frombuffer(b[8:8 + k * LOCUS_ROW.itemsize].tobytes(), dtype=LOCUS_ROW))
    allrows = np.concatenate(parts) if parts else rows
    return np.sort(allrows, order="locus_id")


def packed_words(lengths):
    """words of the packed form per sequence: (2-bit words, mask words) -- every sequence starts on a 64-base boundary"""
    blocks = (np.asarray(lengths, np.int64) + 63) // 64
    return blocks * 4, blocks * 2


def exchange_window_reads_packed(locus_id, read_id, dest, lengths, gather_packed, dist=None, device=None, timings=None):
    """The stage-1 -> per-locus hand-off when reads are sharded over ranks, without ever leaving the device or unpacking a base (round 4;
    the ASCII form of rounds 1-3 was deleted in round 5; the reference does this through the shared file system: pysam.fetch on the
    stage-1 BAM + seqtk over the read file, TELR_assembly.py:384-462): the (locus, read) pairs are put in
    destination order, `gather_packed(order)` hands back the packed words of the reads in that order as two torch int32 tensors
    (the product passes SeqSet.subset(...).packed(): one gather kernel over the resident 2-bit read set; 3 bits per base on the
    wire instead of 8), and TWO collectives move them: the per-peer counts (pairs, words), then ONE all-to-all of int32 words,
    per destination [3 n header words: locus, read, length][2-bit words][mask words].  Nothing is staged through host memory;
    the only host work is the argsort of the (small) pair list.
    -> (locus ids, read ids, lengths) of the received pairs in RECEIVED order (peer by peer), the two word tensors of exactly
    these reads end to end (= the packed form of a set with these lengths: SeqSet.from_packed), and `order` = the permutation
    that sorts the received pairs by (locus id, read id)."""
    import time
    import torch
    t0 = time.time()
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    locus_id = np.asarray(locus_id, np.int64); read_id = np.asarray(read_id, np.int64); dest = np.asarray(dest, np.int64)
    lengths = np.asarray(lengths, np.int64)
    order_s = np.argsort(dest, kind="stable")
    seq2, nmask = gather_packed(order_s)
    dev = seq2.device
    w2, wn = packed_words(lengths[order_s])
    n_to = np.bincount(dest, minlength=world).astype(np.int64)
    cuts = np.concatenate([[0], np.cumsum(n_to)])
    w2_to = np.array([int(w2[cuts[d]:cuts[d + 1]].sum()) for d in range(world)], np.int64)
    hdr = torch.from_numpy(np.stack([locus_id[order_s], read_id[order_s], lengths[order_s]], axis=1).astype(np.int32).reshape(-1)).to(dev)
    if timings is not None:
        timings["pack_s"] = timings.get("pack_s", 0.0) + time.time() - t0
    t0 = time.time()
    if world == 1:
        n_from, w2_from = n_to, w2_to
        got_hdr, got2, gotn = hdr, seq2, nmask
    else:
        # the wire: the device the process group works on (RCCL: the GPU the words are on; the gloo smoke runs of a 1-GPU box and
        # the CPU tests: host memory)
        wire = torch.device(device) if device is not None else torch.device("cpu")
        sizes = torch.from_numpy(np.stack([n_to, w2_to], axis=1).reshape(-1)).to(wire)
        rs = torch.empty_like(sizes)
        dist.all_to_all_single(rs, sizes)
        rs = rs.cpu().numpy().reshape(world, 2)
        n_from, w2_from = rs[:, 0], rs[:, 1]
        c2 = np.concatenate([[0], np.cumsum(w2_to)]); cn = c2 // 2
        send = torch.cat([x for d in range(world) for x in (hdr[3 * cuts[d]:3 * cuts[d + 1]], seq2[c2[d]:c2[d + 1]], nmask[cn[d]:cn[d + 1]])]) if len(dest) else torch.zeros(0, dtype=torch.int32, device=dev)
        in_split = [int(3 * n_to[d] + w2_to[d] + w2_to[d] // 2) for d in range(world)]
        out_split = [int(3 * n_from[p] + w2_from[p] + w2_from[p] // 2) for p in range(world)]
        recv = torch.empty(sum(out_split), dtype=torch.int32, device=wire)
        dist.all_to_all_single(recv, send.to(wire), output_split_sizes=out_split, input_split_sizes=in_split)
        recv = recv.to(dev)
        hs, s2, sn, o = [], [], [], 0
        for p_ in range(world):
            a, b, c = 3 * int(n_from[p_]), int(w2_from[p_]), int(w2_from[p_]) // 2
            hs.append(recv[o:o + a]); s2.append(recv[o + a:o + a + b]); sn.append(recv[o + a + b:o + a + b + c]); o += a + b + c
        got_hdr, got2, gotn = torch.cat(hs), torch.cat(s2), torch.cat(sn)
    h = got_hdr.cpu().numpy().reshape(-1, 3).astype(np.int64)
    if timings is not None:
        timings["collective_s"] = timings.get("collective_s", 0.0) + time.time() - t0
    order = np.lexsort((h[:, 1], h[:, 0]))
    return h[:, 0], h[:, 1], h[:, 2].astype(np.int32), got2, gotn, order


# ---- stage 1 -> ONE coordinate-sorted BAM written by all ranks (round 4) ---------------------------------------------------
# Sniffles reads one file (TELR_sv.py:35-47); the reads were dealt to the ranks.  Instead of bringing everything to rank 0
# (gather_stage1: one writer behind N mappers), the job's records are RANGE-PARTITIONED by coordinate: sampled splitters (one
# small all-gather), every record goes to the rank that owns its coordinate slice together with its read -- as packed 2-bit
# words, device tensors, never ASCII -- and with the OTHER records of that read (marked "not yours": the SA tag of a record
# names its read's other alignments), ONE all-to-all of a byte payload per peer; every rank then codes the BGZF blocks of its
# slice on its own device (telr_write_bam_slice), the slice sizes are scanned over the ranks (one tiny all-gather), every rank
# puts its image at its place of the one file, and rank 0 writes the one .bai from the records' coordinates and virtual offsets.


def _pad8(n):
    return (int(n) + 7) & ~7


def stage1_splitters(keys, world, dist, wire, per_rank=64):
    """coordinate keys (int64, this rank's records) -> world - 1 ascending splitters, the same on every rank: the quantiles of
    per_rank * world samples of every rank's sorted keys (ONE all-gather of fixed-size blocks)"""
    import torch
    n = per_rank * world
    ks = np.sort(np.asarray(keys, np.int64))
    samp = ks[(np.arange(n) * len(ks)) // n] if len(ks) else np.full(n, np.iinfo(np.int64).max, np.int64)
    mine = torch.from_numpy(np.ascontiguousarray(samp)).to(wire)
    got = torch.empty(world * n, dtype=torch.int64, device=wire)
    dist.all_gather_into_tensor(got, mine)
    allk = np.sort(got.cpu().numpy())
    allk = allk[allk != np.iinfo(np.int64).max]
    if len(allk) == 0:
        return np.zeros(world - 1, np.int64)
    return allk[(np.arange(1, world) * len(allk)) // world]


def exchange_stage1(alns, cig_t, lengths, names, gid, gather_packed, rec_dest, world, dist, wire, timings=None):
    """Every record (with its read and the read's other records) to the rank `rec_dest` says; reads without a record to the last
    rank.  alns: this rank's records, query-major, qid = index into this rank's reads (lengths / names / gid, ascending gid);
    cig_t: the CIGAR words as a torch int32 tensor (device); gather_packed(read indices) -> the packed words of those reads.
    -> dict(alns, emit, cig (torch int32), lengths, names, gid, seq2, nmask) of what this rank now holds: reads in ascending gid
    order, records query-major with qid / cigar_off re-based, emit[i] = record i lies in this rank's slice."""
    import time
    import torch
    t0 = time.time()
    err = None
    try:
        rank = dist.get_rank()
        alns = np.ascontiguousarray(alns)
        nq = len(lengths)
        lengths = np.asarray(lengths, np.int32); gid = np.asarray(gid, np.int64)
        qid = alns["qid"].astype(np.int64)
        rec_dest = np.asarray(rec_dest, np.int64)
        rstart = np.searchsorted(qid, np.arange(nq)); rcount = np.searchsorted(qid, np.arange(nq), side="right") - rstart
        # (destination, read) pairs: every slice one of the read's records lies in; reads without records -> the last rank
        pair = np.unique(np.concatenate([rec_dest * nq + qid, (world - 1) * nq + np.nonzero(rcount == 0)[0]]))       # sorted by (dest, read) = (dest, gid)
        p_dest = pair // nq; p_read = pair % nq
        n_to = np.bincount(p_dest, minlength=world).astype(np.int64)
        cuts = np.concatenate([[0], np.cumsum(n_to)])
        # the records that travel with every pair
        p_nrec = rcount[p_read]
        tot_rec = int(p_nrec.sum())
        pr_off = np.cumsum(p_nrec) - p_nrec
        ridx = np.repeat(rstart[p_read] - pr_off, p_nrec) + np.arange(tot_rec, dtype=np.int64)          # record index per sent record
        r_pair = np.repeat(np.arange(len(pair)), p_nrec)
        s_alns = alns[ridx].copy()
        s_emit = (rec_dest[ridx] == p_dest[r_pair]).astype(np.uint8)
        s_alns["qid"] = (r_pair - cuts[p_dest[r_pair]]).astype(np.int32)                                # index of the read inside its destination block
        rec_cuts = np.concatenate([[0], np.cumsum(np.bincount(p_dest[r_pair], minlength=world))]).astype(np.int64)
        ncig = s_alns["n_cigar"].astype(np.int64)
        c_off_all = np.cumsum(ncig) - ncig
        cig_cuts = np.concatenate([[0], np.cumsum(np.bincount(p_dest[r_pair], weights=ncig, minlength=world))]).astype(np.int64)       # CIGAR words per destination block
        old_off = alns["cigar_off"].astype(np.int64)[ridx]
        s_alns["cigar_off"] = (c_off_all - cig_cuts[p_dest[r_pair]]).astype(s_alns["cigar_off"].dtype) if tot_rec else s_alns["cigar_off"]
        dev = cig_t.device
        _sub(timings, "pack_host_index_s", t0, dev); t1 = time.time()
        # the CIGAR words of the sent records, gathered on the device
        tot_cig = int(ncig.sum())
        if tot_cig:
            src = torch.from_numpy(np.ascontiguousarray(old_off - c_off_all)).to(dev)
            widx = torch.repeat_interleave(src, torch.from_numpy(ncig).to(dev)) + torch.arange(tot_cig, device=dev, dtype=torch.int64)
            s_cig = cig_t.view(torch.int32)[widx]
        else:
            s_cig = torch.zeros(0, dtype=torch.int32, device=dev)
        _sub(timings, "pack_cigar_gather_s", t1, dev); t1 = time.time()
        seq2, nmask = gather_packed(p_read)
        _sub(timings, "pack_read_subset_s", t1, dev)
        w2, _ = packed_words(lengths[p_read])
        w2_cuts = np.concatenate([[0], np.cumsum(w2)])[cuts]
        _sub(timings, "pack_s", t0, dev)
        t0 = time.time()
        # per destination: [records][emit][gid][length][names][CIGAR words][2-bit words][mask words], every section padded to 8 bytes
        sec = np.zeros((world, 8), np.int64)
        parts = []
        for d in range(world):
            a = s_alns[rec_cuts[d]:rec_cuts[d + 1]]; e = s_emit[rec_cuts[d]:rec_cuts[d + 1]]
            rd = p_read[cuts[d]:cuts[d + 1]]
            nb = "\n".join(names[i] for i in rd).encode()
            host = [a.view(np.uint8).reshape(-1), e, gid[rd].view(np.uint8).reshape(-1), lengths[rd].view(np.uint8).reshape(-1), np.frombuffer(nb, np.uint8)]
            devp = [s_cig[cig_cuts[d]:cig_cuts[d + 1]], seq2[w2_cuts[d]:w2_cuts[d + 1]], nmask[w2_cuts[d] // 2:w2_cuts[d + 1] // 2]]
            sec[d, :5] = [len(x) for x in host]; sec[d, 5:] = [int(x.numel()) * 4 for x in devp]
            hb = np.zeros(sum(_pad8(len(x)) for x in host), np.uint8); o = 0
            for x in host:
                hb[o:o + len(x)] = x; o += _pad8(len(x))
            parts.append(torch.from_numpy(hb).to(dev))
            for x in devp:
                parts.append(x.contiguous().view(torch.uint8))
                if (x.numel() * 4) % 8:
                    parts.append(torch.zeros(4, dtype=torch.uint8, device=dev))
        send = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.uint8, device=dev)
        _sub(timings, "payload_build_s", t0, dev)
    except Exception as e:          # (out of memory while packing, a bad index ...): the peers must hear of it before they enter the all-to-all
        err = e
    _agree(dist, wire, err, "packing the records and reads")
    in_split = [int(sum(_pad8(v) for v in sec[d])) for d in range(world)]
    st = torch.from_numpy(sec.reshape(-1).copy()).to(wire); rs = torch.empty_like(st)
    dist.all_to_all_single(rs, st)
    rsec = rs.cpu().numpy().reshape(world, 8)
    out_split = [int(sum(_pad8(v) for v in rsec[p_])) for p_ in range(world)]
    # the largest allocation of the exchange (the receive buffer) and the host / device staging of the payload: a rank that runs out of
    # memory HERE must take every rank out before the payload all-to-all, not leave its peers waiting in it
    err = None; recv = send_w = None
    try:
        recv = torch.empty(sum(out_split), dtype=torch.uint8, device=wire)
        send_w = send.to(wire)
    except Exception as e:
        err = e
    _agree(dist, wire, err, "allocating the receive buffer of the record exchange")
    dist.all_to_all_single(recv, send_w, output_split_sizes=out_split, input_split_sizes=in_split)
    del send_w
    err = None
    try:
        recv = recv.to(dev)
    except Exception as e:
        err = e
    _agree(dist, wire, err, "moving the received records to the device")
    if timings is not None:
        timings["collective_s"] = timings.get("collective_s", 0.0) + time.time() - t0
        timings["bytes_sent"] = int(sum(in_split)) - in_split[rank]
        # what the payload travelled on (round 6): under RCCL the wire is the device the words already lie on -- no host
        # staging on either side -- and at world size 1 the rank's whole payload goes through the all-to-all to itself
        timings["wire"] = str(wire); timings["payload_sent_from"] = str(send.device); timings["payload_received_on"] = str(recv.device)
        timings["bytes_to_self"] = int(in_split[rank])
    t0 = time.time()
    err = None
    try:
        g_alns, g_emit, g_gid, g_len, g_names, g_cig, g_s2, g_sn = [], [], [], [], [], [], [], []
        o = 0; q0 = 0; c0 = 0
        for p_ in range(world):
            v = rsec[p_]; pos = [o]
            for x in v:
                pos.append(pos[-1] + _pad8(x))
            host = recv[pos[0]:pos[5]].cpu().numpy()
            h0 = pos[0]
            a = np.frombuffer(host[pos[0] - h0:pos[0] - h0 + v[0]].tobytes(), dtype=alns.dtype).copy()
            a["qid"] += q0; a["cigar_off"] += c0
            ln = np.frombuffer(host[pos[3] - h0:pos[3] - h0 + v[3]].tobytes(), np.int32)
            g_alns.append(a); g_emit.append(host[pos[1] - h0:pos[1] - h0 + v[1]].copy())
            g_gid.append(np.frombuffer(host[pos[2] - h0:pos[2] - h0 + v[2]].tobytes(), np.int64)); g_len.append(ln)
            g_names += host[pos[4] - h0:pos[4] - h0 + v[4]].tobytes().decode().split("\n") if len(ln) else []
            g_cig.append(recv[pos[5]:pos[5] + v[5]].view(torch.int32)); g_s2.append(recv[pos[6]:pos[6] + v[6]].view(torch.int32)); g_sn.append(recv[pos[7]:pos[7] + v[7]].view(torch.int32))
            q0 += len(ln); c0 += int(v[5]) // 4
            o = pos[8]
        a_all = np.concatenate(g_alns); e_all = np.concatenate(g_emit); gid_all = np.concatenate(g_gid); len_all = np.concatenate(g_len)
        cig_all = torch.cat(g_cig); s2 = torch.cat(g_s2); sn = torch.cat(g_sn)
        _sub(timings, "unpack_parse_s", t0, dev)
        t1 = time.time()
        # reads into ascending job order (ties of the coordinate sort are broken by it): a permutation of the packed pieces on the device
        order = np.argsort(gid_all, kind="stable")
        if len(order) and not (order == np.arange(len(order))).all():
            w2a, _ = packed_words(len_all)
            st2 = np.cumsum(w2a) - w2a
            src = torch.from_numpy(np.ascontiguousarray(st2[order] - (np.cumsum(w2a[order]) - w2a[order]))).to(dev)
            cnt = torch.from_numpy(np.ascontiguousarray(w2a[order])).to(dev)
            tot2 = int(w2a.sum())
            i2 = torch.repeat_interleave(src, cnt) + torch.arange(tot2, device=dev, dtype=torch.int64)
            s2 = s2[i2]
            half = torch.repeat_interleave(src // 2, cnt // 2) + torch.arange(tot2 // 2, device=dev, dtype=torch.int64)
            sn = sn[half]
            place = np.empty(len(order), np.int64); place[order] = np.arange(len(order))
            a_all["qid"] = place[a_all["qid"]]
            ro = np.argsort(a_all["qid"], kind="stable")
            a_all = a_all[ro]; e_all = e_all[ro]
            len_all = len_all[order]; gid_all = gid_all[order]; g_names = [g_names[i] for i in order]
        _sub(timings, "unpack_reorder_s", t1, dev)
        if timings is not None:
            timings["unpack_s"] = timings.get("unpack_s", 0.0) + time.time() - t0
    except Exception as e:          # (no room for the re-ordered copies, ...)
        err = e
    _agree(dist, wire, err, "unpacking the received records and reads")
    return dict(alns=a_all, emit=e_all, cig=cig_all, lengths=len_all, names=g_names, gid=gid_all, seq2=s2, nmask=sn)


class JobBamError(RuntimeError):
    """raised on EVERY rank when any rank failed inside a collective section of write_job_bam"""


def _agree(dist, wire, err, what):
    """One all-reduce(MAX) of an ok / failed flag: every rank leaves here the same way.  A rank that fails between two collectives
    must not simply unwind -- its peers would block in the next all-gather / all-to-all for ever."""
    import torch
    f = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=wire)
    dist.all_reduce(f, op=dist.ReduceOp.MAX)
    if int(f.item()):
        raise JobBamError("write_job_bam: %s failed on %s" % (what, "this rank: %s: %s" % (type(err).__name__, err) if err is not None else "another rank"))


def _sub(timings, key, t0, dev=None):
    """sub-phase timer; with TELR_PHASE_SYNC=1 the device is drained first, so that a phase is charged with its own kernels and
    copies instead of whatever was still queued when its first blocking call came"""
    import os, time
    if timings is None:
        return
    if os.environ.get("TELR_PHASE_SYNC") and dev is not None and getattr(dev, "type", "") == "cuda":
        import torch
        torch.cuda.synchronize(dev)
    timings[key] = timings.get(key, 0.0) + time.time() - t0


def write_job_bam(path, ix, eng, alns, cigars, read_set, lengths, names, gid, tnames, tlens, dist, device=None, level=1, writer_kw=None, timings=None):
    """ONE coordinate-sorted BAM + .bai for the job, written by all ranks (see the block comment above).  Collective: every rank
    calls it with its own stage-1 result (alns, cigars: host arrays of this rank's telr_map), its resident read set (SeqSet) and
    the reads' lengths / names / job-level numbers (ascending).  tnames / tlens: the reference.  -> on every rank a dict of
    phase seconds and sizes (rank 0: the job's totals as well)."""
    import time
    import torch
    from .aligner import SeqSet
    tm = {} if timings is None else timings
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device("cuda", eng.device)
    wire = torch.device(device) if device is not None else torch.device("cpu")
    kw = dict(md=True, cs=True, softclip=True, cmdline="telr_map"); kw.update(writer_kw or {})
    t0 = time.time()
    alns = np.ascontiguousarray(alns)
    # room for the wire buffers (torch allocates them next to the engine): CIGAR words and packed reads exist up to ~6 times on
    # their way (source, gathered, payload, received, re-ordered, the writer's copies); the context's grow-only mapping scratch
    # (150+ GB after a 30x stage 1) goes back first when the device is that full -- the next telr_map sizes it again
    payload = 4 * len(cigars) + int(np.asarray(lengths, np.int64).sum()) * 3 // 8 + alns.nbytes
    fr, _tot = eng.mem_info()
    if fr < 8 * payload + (2 << 30):
        t1 = time.time()
        eng.release_scratch()
        _sub(tm, "partition_release_scratch_s", t1, dev)
    t1 = time.time()
    keys = alns["tid"].astype(np.int64) << 32 | alns["ts"].astype(np.int64)
    split = stage1_splitters(keys, world, dist, wire)
    dest = np.searchsorted(split, keys, side="right")
    _sub(tm, "partition_splitters_s", t1, dev); t1 = time.time()
    held = []
    seg = jr = jq = None
    ok = False; created = False
    try:
        err = None; cig_t = None
        try:
            cig_t = torch.from_numpy(np.ascontiguousarray(cigars, dtype=np.uint32).view(np.int32)).to(dev)
        except Exception as e:
            err = e
        _agree(dist, wire, err, "uploading the CIGAR words")
        _sub(tm, "partition_cigar_upload_s", t1, dev)

        def gather_packed(idx):
            sub = read_set.subset(np.asarray(idx, np.int32)); held.append(sub)
            return sub.packed()
        tm["partition_s"] = time.time() - t0
        got = exchange_stage1(alns, cig_t, lengths, names, gid, gather_packed, dest, world, dist, wire, timings=tm)
        while held:
            held.pop().free()
        del cig_t
        t0 = time.time()
        err = None; info = None
        try:
            jq = SeqSet.from_packed(eng, got["lengths"], got["seq2"], got["nmask"])
            _sub(tm, "slice_from_packed_s", t0, dev); t1 = time.time()
            jr = ix.result_from_device_cigars(got["alns"], got["cig"])
            _sub(tm, "slice_result_s", t1, dev); t1 = time.time()
            got["seq2"] = got["nmask"] = got["cig"] = None          # the library holds its own copies now: torch's cache goes back to the device for the writer
            if dev.type == "cuda":
                torch.cuda.empty_cache()
            seg = ix.write_bam_slice(jr, jq, ix._cstr_array(got["names"]), tnames, got["emit"], with_header=rank == 0, unmapped=rank == world - 1, level=level, **kw)
            info = ix.segment_info(seg)
            _sub(tm, "slice_code_s", t1, dev)
        except Exception as e:              # out of device memory in the coder, ...
            err = e
        _agree(dist, wire, err, "coding the BGZF blocks of a slice")
        tm["code_slice_s"] = time.time() - t0
        t0 = time.time()
        sizes = torch.zeros(world, dtype=torch.int64, device=wire); mine = torch.tensor([info["bytes"]], dtype=torch.int64, device=wire)
        dist.all_gather_into_tensor(sizes, mine)
        sizes = sizes.cpu().numpy()
        base = int(sizes[:rank].sum()); total = int(sizes.sum()) + 28
        err = None
        if rank == 0:                                   # the file exists at its final length before anybody writes into it
            try:
                with open(path, "wb") as fh:
                    fh.truncate(total)
            except Exception as e:
                err = e
        _agree(dist, wire, err, "creating the job's file")          # (also the barrier: nobody writes before the file exists)
        created = True                                              # from here on a failed call removes the file; before, `path` may hold an earlier run's BAM
        tm["scan_sizes_s"] = time.time() - t0
        t0 = time.time()
        err = None
        try:
            ix.segment_write(seg, path, base, rank == world - 1)
        except Exception as e:              # ENOSPC, a vanished directory, ...
            err = e
        _agree(dist, wire, err, "writing a slice into the job's file")
        tm["write_slice_s"] = time.time() - t0
        t0 = time.time()
        err = None; ent = np.zeros(0, np.int64)
        try:
            tid, ts, te, vb, v_end = ix.segment_entries(seg, base)
            ent = np.concatenate([tid.astype(np.int64), ts.astype(np.int64), te.astype(np.int64), vb.view(np.int64), np.array([v_end, info["unmapped_reads"]], np.uint64).view(np.int64)])
        except Exception as e:
            err = e
        _agree(dist, wire, err, "collecting a slice's index entries")
        n_ent = torch.zeros(world, dtype=torch.int64, device=wire)
        dist.all_gather_into_tensor(n_ent, torch.tensor([len(ent)], dtype=torch.int64, device=wire))
        n_ent = [int(x) for x in n_ent.cpu().numpy()]
        err = None; recv = ent_w = None
        try:
            recv = torch.empty(sum(n_ent) if rank == 0 else 0, dtype=torch.int64, device=wire)
            ent_w = torch.from_numpy(ent).to(wire)
        except Exception as e:
            err = e
        _agree(dist, wire, err, "allocating the index entries' exchange")
        dist.all_to_all_single(recv, ent_w, output_split_sizes=n_ent if rank == 0 else [0] * world, input_split_sizes=[len(ent)] + [0] * (world - 1))
        out = dict(tm, slice_bytes=info["bytes"], slice_records=info["mapped_records"], slice_unmapped_reads=info["unmapped_reads"], reads_held=int(len(got["lengths"])),
                   records_held=int(len(got["alns"])))
        err = None
        if rank == 0:
            try:
                raw = recv.cpu().numpy(); o = 0
                T, S, E, V = [], [], [], []
                v_last, n_un = 0, 0
                for r_ in range(world):
                    n = (n_ent[r_] - 2) // 4
                    b = raw[o:o + n_ent[r_]]; o += n_ent[r_]
                    T.append(b[:n]); S.append(b[n:2 * n]); E.append(b[2 * n:3 * n]); V.append(b[3 * n:4 * n].view(np.uint64))
                    v_last = int(b[4 * n:].view(np.uint64)[0]); n_un += int(b[4 * n + 1])
                ix.bai_write(path + ".bai", np.concatenate(T), np.concatenate(S), np.concatenate(E), np.concatenate(V), v_last, n_un, tlens)
                out.update(bam_bytes=total, records=int(sum(len(x) for x in T)), unmapped_reads=n_un)
            except Exception as e:
                err = e
        out["index_s"] = time.time() - t0
        _agree(dist, wire, err, "writing the index")                 # (the closing barrier of the call)
        ok = True
        return out
    finally:
        # whatever happened: the slice, the result, the received read set and the gathered subsets go back to the device, and a
        # failed call leaves no file behind (rank 0 created it)
        for sub in held:
            try:
                sub.free()
            except Exception:
                pass
        if seg is not None:
            ix.segment_free(seg)
        if jr is not None:
            ix.free_raw(jr)
        if jq is not None:
            jq.free()
        if not ok and created and rank == 0:
            import os
            for f in (path, path + ".bai"):
                try:
                    os.unlink(f)
                except OSError:
                    pass


def rows_from_reports(locus_ids, reports, freqs, chrom_ids, family_ids):
    """liftover report dicts (+ te_freq dicts) -> LOCUS_ROW array"""
    out = np.zeros(len(locus_ids), LOCUS_ROW)
    for k, (lid, rep, fr) in enumerate(zip(locus_ids, reports, freqs)):
        r = rep["report"]
        o = out[k]
        o["locus_id"] = lid; o["status"] = rep["num_hits"]; o["type"] = TYPE_CODES[r["type"]]
        o["chrom_id"] = chrom_ids.get(r["chrom"], -1) if r["chrom"] is not None else -1
        for name in ("start", "end", "gap"):
            o[name] = NONE_I32 if r.get(name) is None else r[name]
        o["tsd_len"] = NONE_I32 if r.get("TSD_length") is None else r["TSD_length"]
        o["strand"] = 0 if r["strand"] is None else (1 if r["strand"] == "+" else -1)
        fams = [family_ids[f] for f in str(r["family"]).split("|")][:4]
        o["n_family"] = len(fams); o["family_id"][:len(fams)] = fams
        if fr is not None:
            keys = ("te_5p_cov", "te_3p_cov", "flank_5p_cov", "flank_3p_cov", "te_5p_cov_rc", "te_3p_cov_rc", "flank_5p_cov_rc", "flank_3p_cov_rc")
            o["medians"] = [np.nan if fr.get(x) is None else fr[x] for x in keys]
            o["af"] = np.nan if fr.get("freq") is None else fr["freq"]
        else:
            o["medians"] = np.nan; o["af"] = np.nan
    return out
