"""Sharding of the path across the GPUs of one node.

* stage 1: reads are dealt to ranks by cumulative bases (length-sorted, snake order) against a
  replicated index — no collective on the data path;
* stages 2-4: candidate loci are assigned by LPT (longest processing time first) on their contig
  + read bases; every rank produces fixed-width result rows for its loci and ONE all-gather of
  fixed-capacity blocks merges them (RCCL over xGMI on GPUs, `backend="nccl"`; gloo in the CPU tests).
  Variable-length payloads (sequences, CIGARs) stay on the owning rank;
* between the two (only when the reads themselves are sharded): the window reads of a locus sit on whichever
  ranks mapped them and travel to the locus' owner in one all-to-all (`exchange_window_reads`).
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


READ_HDR = np.dtype([("locus_id", np.int32), ("read_id", np.int32), ("length", np.int32), ("pad", np.int32)])


def exchange_window_reads(locus_id, read_id, dest, reads, read_index, dist=None, device=None):
    """The stage-1 -> per-locus hand-off when reads are sharded over ranks: every rank holds the window reads of ALL loci
    that fall in ITS read shard and sends each to the rank that owns the locus.  (The reference does this through the
    shared file system: pysam.fetch on the stage-1 BAM + seqtk over the read file, TELR_assembly.py:384-462.)

    One entry per (locus, read) pair to send: locus_id[i], read_id[i] (global read id), dest[i] (owner rank of the locus),
    read_index[i] = index of the read in `reads` = (buf, off, len), this rank's read set on the host.
    -> (locus ids, read ids, (buf, off, len)) of the pairs this rank received, sorted by (locus id, read id).
    Two all-to-all collectives: the byte counts, then one buffer per peer [n][n x READ_HDR][bases]."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    locus_id = np.asarray(locus_id, np.int64); read_id = np.asarray(read_id, np.int64); dest = np.asarray(dest, np.int64)
    read_index = np.asarray(read_index, np.int64)
    rbuf, roff, rln = reads
    roff = np.asarray(roff, np.int64); rln = np.asarray(rln, np.int64)

    def pack(sel):
        """[n][headers][bases] of the entries `sel`"""
        hdr = np.zeros(len(sel), READ_HDR)
        hdr["locus_id"] = locus_id[sel]; hdr["read_id"] = read_id[sel]; hdr["length"] = rln[read_index[sel]]
        # the bases of all selected reads in one gather: flat index = start of the read, repeated, + position inside it
        ri = read_index[sel]; ln = rln[ri]
        tot = int(ln.sum())
        if tot:
            ends = np.cumsum(ln)
            flat = np.repeat(roff[ri] - (ends - ln), ln) + np.arange(tot, dtype=np.int64)
            bases = np.asarray(rbuf)[flat]
        else:
            bases = np.zeros(0, np.uint8)
        return np.concatenate([np.frombuffer(np.int64(len(sel)).tobytes(), np.uint8), hdr.view(np.uint8).reshape(-1), bases])

    def unpack(raw, sizes):
        """-> header array and base offsets (into raw) of all reads of the concatenated per-peer buffers"""
        hdrs, offs, o = [], [], 0
        for sz in sizes:
            b = raw[o:o + sz]
            n = int(np.frombuffer(b[:8].tobytes(), np.int64)[0]) if sz >= 8 else 0
            h = np.frombuffer(b[8:8 + n * READ_HDR.itemsize].tobytes(), READ_HDR)
            start = o + 8 + n * READ_HDR.itemsize
            ln = h["length"].astype(np.int64)
            hdrs.append(h); offs.append(start + np.cumsum(ln) - ln)
            o += sz
        h = np.concatenate(hdrs) if hdrs else np.zeros(0, READ_HDR)
        return h, (np.concatenate(offs) if offs else np.zeros(0, np.int64))

    if world == 1:
        raw = pack(np.arange(len(dest))); sizes = [len(raw)]
    else:
        import torch
        bufs = [pack(np.nonzero(dest == d)[0]) for d in range(world)]
        dev = device if device is not None else "cpu"
        n_send = torch.tensor([len(b) for b in bufs], dtype=torch.int64, device=dev)
        n_recv = torch.empty(world, dtype=torch.int64, device=dev)
        dist.all_to_all_single(n_recv, n_send)
        sizes = [int(x) for x in n_recv.cpu().tolist()]
        send = torch.from_numpy(np.concatenate(bufs)).to(dev)
        recv = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
        dist.all_to_all_single(recv, send, output_split_sizes=sizes, input_split_sizes=[len(b) for b in bufs])
        raw = recv.cpu().numpy()
    h, off = unpack(raw, sizes)
    order = np.lexsort((h["read_id"], h["locus_id"]))
    return h["locus_id"][order].astype(np.int64), h["read_id"][order].astype(np.int64), (raw, off[order], h["length"][order].astype(np.int32))


def packed_words(lengths):
    """words of the packed form per sequence: (2-bit words, mask words) -- every sequence starts on a 64-base boundary"""
    blocks = (np.asarray(lengths, np.int64) + 63) // 64
    return blocks * 4, blocks * 2


def exchange_window_reads_packed(locus_id, read_id, dest, lengths, gather_packed, dist=None, device=None, timings=None):
    """`exchange_window_reads` without ever leaving the device or unpacking a base (round 4): the (locus, read) pairs are put in
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


def gather_stage1(alns, cigars, reads, read_names, dist=None, device=None, force=False, read_gid=None):
    """The stage-1 hand-off at N > 1: Sniffles wants ONE coordinate-sorted BAM, the reads were dealt to the ranks.  Every rank
    packs what it mapped -- records, CIGAR words, read bases, lengths and names -- into one byte blob; ONE all-gather of the five
    sizes and ONE all-to-all whose only non-empty destination is rank 0 (RCCL over xGMI on device tensors, gloo in the tests: the
    same two collectives the loci leg uses) bring them to rank 0, which re-bases the query ids and CIGAR offsets and holds the job's
    records in rank order: the input of ONE telr_write_bam_dev call ("each rank writes its own shard, host merges" --
    the merge is the device writer's sort; writing is bound by the host's page cache, so a second writer would not help).
    force: run the collectives at world size 1 too.  read_gid: the job-level number of every read of this rank (its place in the
    input file); with it rank 0 puts reads and records back into file order, so the job's arrays -- and the BAM written from them,
    ties in the coordinate sort included -- do not depend on how the reads were dealt.
    -> on rank 0: (alns, cigars, (buf, off, len), names) of the whole job (with read_gid the offsets are not ascending: read i of the job
    is buf[off[i] : off[i] + len[i]]); on the other ranks None."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    buf, off, ln = reads
    alns = np.ascontiguousarray(alns); cigars = np.ascontiguousarray(cigars, dtype=np.uint32)
    ln = np.ascontiguousarray(ln, np.int32); off = np.asarray(off, np.int64)
    if world == 1 and not (force and dist is not None and dist.is_initialized()):
        return alns, cigars, (buf, off, ln), list(read_names)
    import torch
    dev = device if device is not None else "cpu"
    names_blob = np.frombuffer(("\n".join(read_names)).encode(), np.uint8)
    # the reads of this rank end to end (they usually are already)
    if len(ln) and not (off == np.cumsum(ln.astype(np.int64)) - ln).all():
        buf = np.concatenate([buf[o:o + l] for o, l in zip(off, ln)])
    gid = np.zeros(0, np.int64) if read_gid is None else np.ascontiguousarray(read_gid, np.int64)
    mine = [alns.view(np.uint8).reshape(-1), cigars.view(np.uint8).reshape(-1), np.ascontiguousarray(buf, np.uint8)[:int(ln.sum())], ln.view(np.uint8).reshape(-1), names_blob,
            gid.view(np.uint8).reshape(-1)]
    sizes = torch.tensor([len(x) for x in mine], dtype=torch.int64, device=dev)
    all_sizes = [torch.empty(6, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    all_sizes = [[int(v) for v in t.cpu().tolist()] for t in all_sizes]
    blob = torch.from_numpy(np.concatenate(mine)).to(dev)
    recv_n = [sum(x) for x in all_sizes] if rank == 0 else [0] * world
    got = torch.empty(sum(recv_n), dtype=torch.uint8, device=dev)
    dist.all_to_all_single(got, blob, output_split_sizes=recv_n, input_split_sizes=[len(blob)] + [0] * (world - 1))
    if rank != 0:
        return None
    got = got.cpu().numpy()
    out_alns, out_cig, out_buf, out_len, out_names, out_gid = [], [], [], [], [], []
    q0 = 0; c0 = 0; p0 = 0
    for r in range(world):
        p = []
        for k in range(6):
            p.append(got[p0:p0 + all_sizes[r][k]]); p0 += all_sizes[r][k]
        a = np.frombuffer(p[0].tobytes(), dtype=alns.dtype).copy()
        a["qid"] += q0; a["cigar_off"] += c0
        lens = np.frombuffer(p[3].tobytes(), np.int32)
        out_alns.append(a); out_cig.append(np.frombuffer(p[1].tobytes(), np.uint32)); out_buf.append(p[2]); out_len.append(lens)
        out_names += p[4].tobytes().decode().split("\n") if len(lens) else []
        out_gid.append(np.frombuffer(p[5].tobytes(), np.int64))
        q0 += len(lens); c0 += len(p[1]) // 4
    ln_all = np.concatenate(out_len); a_all = np.concatenate(out_alns); buf_all = np.concatenate(out_buf); gid_all = np.concatenate(out_gid)
    off_all = np.cumsum(ln_all.astype(np.int64)) - ln_all
    if read_gid is not None:
        if len(gid_all) != len(ln_all):
            raise ValueError("gather_stage1: read_gid must be given by every rank or by none")
        order = np.argsort(gid_all, kind="stable")                  # new place -> gathered place
        place = np.empty(len(order), np.int64); place[order] = np.arange(len(order))
        a_all["qid"] = place[a_all["qid"]]
        a_all = a_all[np.argsort(a_all["qid"], kind="stable")]      # the records of a read stay in the order the engine gave them
        # the bases stay where the gather put them: a sequence set is (buffer, offsets, lengths), so putting the reads into file
        # order is a permutation of the two small arrays (moving 4 GB of bases through a gather index would cost 60 GB of host memory)
        ln_all = ln_all[order]; off_all = off_all[order]
        out_names = [out_names[i] for i in order]
    return a_all, np.concatenate(out_cig), (buf_all, off_all, ln_all), out_names


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
