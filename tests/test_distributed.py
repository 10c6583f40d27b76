"""N>1 path on CPU: world_size-2 gloo processes shard loci, compute their rows and merge them with the
single all-gather; the merged table must equal the world_size-1 result."""
import json
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make_rows(ids):
    from telr_amd import shard
    rows = np.zeros(len(ids), shard.LOCUS_ROW)
    for k, i in enumerate(ids):
        rng = np.random.default_rng(1000 + i)
        rows[k]["locus_id"] = i; rows[k]["start"] = int(rng.integers(0, 1 << 20)); rows[k]["end"] = rows[k]["start"] + 5
        rows[k]["strand"] = 1 if i % 2 else -1; rows[k]["type"] = 1; rows[k]["af"] = round(float(rng.random()), 3)
        rows[k]["medians"] = rng.integers(0, 40, size=8).astype(np.float32); rows[k]["support"] = int(rng.integers(3, 60))
    return rows


def _worker(rank, world, port, n_loci, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from telr_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    costs = [(i * 7919) % 1000 + 1 for i in range(n_loci)]
    shards = shard.shard_loci(costs, world)
    mine = shards[rank]
    merged = shard.all_gather_rows(_make_rows(mine), dist, capacity=max(len(x) for x in shards))
    if rank == 0:
        np.save(out_path, merged)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_locus_rows_allgather_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    n_loci, world = 37, 2
    out = str(tmp_path / "merged.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_loci, out), nprocs=world, join=True)
    merged = np.load(out)
    want = _make_rows(list(range(n_loci)))
    assert len(merged) == n_loci
    want = np.sort(want, order="locus_id")
    for name in merged.dtype.names:          # field-wise: struct padding bytes are not data
        np.testing.assert_array_equal(merged[name], want[name], err_msg=name)


def test_shard_reads_and_loci_balance():
    from telr_amd import shard
    rng = np.random.default_rng(3)
    lens = rng.lognormal(9, 0.6, size=1001).astype(np.int64)
    for world in (1, 2, 4, 8):
        parts = shard.shard_reads(lens, world)
        assert sorted(sum(parts, [])) == list(range(len(lens)))
        bases = [int(lens[p].sum()) for p in parts]
        assert max(bases) - min(bases) <= 0.03 * sum(bases) / world + lens.max()
        lp = shard.shard_loci(lens[:200], world)
        assert sorted(sum(lp, [])) == list(range(200))
        loads = [int(lens[:200][p].sum()) for p in lp]
        assert max(loads) - min(loads) <= lens[:200].max()


def test_rows_from_reports_roundtrip():
    from telr_amd import shard
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "liftover_driver.json")))
    reps = g["expected_report"]
    fr = [{"te_5p_cov": 13.0, "te_3p_cov": 12.0, "flank_5p_cov": 18.0, "flank_3p_cov": 17.0, "te_5p_cov_rc": 14.0, "te_3p_cov_rc": 15.0,
           "flank_5p_cov_rc": 18.0, "flank_3p_cov_rc": 19.0, "freq": 0.75}] * len(reps)
    rows = shard.rows_from_reports(list(range(len(reps))), reps, fr, {"chr2L": 0}, {"roo": 0, "jockey": 1, "copia": 2})
    assert list(rows["start"]) == [r["report"]["start"] for r in reps]
    assert list(rows["strand"]) == [1 if r["report"]["strand"] == "+" else -1 for r in reps]
    assert rows["af"][0] == 0.75 and rows["tsd_len"][3] == shard.NONE_I32


def _locus_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from telr_amd import locus_pipeline
    from telr_amd.presets import preset
    from oracle_backend import OracleBackend
    from locus_data import make_loci
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref, lib_names, lib, loci, truth = make_loci(n_ins=5, reads_per_locus=16)
    be = OracleBackend(); io, _ = preset("asm10")
    rows, _ = locus_pipeline.run_loci_distributed(be, be.index([ref], io), ["chr2L"], lambda ch: ref, loci, lib_names, lib, dist=dist)
    if rank == 0:
        np.save(out_path, rows)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_locus_bundle_sharded_world2_equals_world1(tmp_path):
    """the per-locus bundle sharded over 2 gloo ranks + ONE all-gather == the single-process table"""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from telr_amd import locus_pipeline
    from telr_amd.presets import preset
    from oracle_backend import OracleBackend
    from locus_data import make_loci
    out = str(tmp_path / "rows.npy")
    mp.spawn(_locus_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    merged = np.load(out)
    ref, lib_names, lib, loci, truth = make_loci(n_ins=5, reads_per_locus=16)
    be = OracleBackend(); io, _ = preset("asm10")
    rows1, _ = locus_pipeline.run_loci_distributed(be, be.index([ref], io), ["chr2L"], lambda ch: ref, loci, lib_names, lib)
    assert len(merged) == len(rows1) >= 4
    for name in merged.dtype.names:
        a, b = merged[name], rows1[name]
        if a.dtype.kind == "f":
            np.testing.assert_array_equal(np.nan_to_num(a, nan=-1.0), np.nan_to_num(b, nan=-1.0), err_msg=name)
        else:
            np.testing.assert_array_equal(a, b, err_msg=name)
    # coordinates of the merged table against the truth
    for r in merged:
        t = truth[int(r["locus_id"])]
        if r["type"] == 1:
            assert abs(int(r["start"]) - t["pos"]) <= 20 and (1 if t["strand"] == "+" else -1) == r["strand"]


def _exchange_items(rank, world):
    """(locus, global read id, destination, local read index) of what `rank` sends; rank 1 sends nothing"""
    out = []
    if rank != 1:
        for k in range(7):
            gid = 100 * rank + k
            for locus in range(5):
                if (gid + locus) % 3 == 0:
                    out.append((locus, gid, locus % world, k))
    return out


def _exchange_reads(rank):
    seqs = [np.random.default_rng(100 * rank + k).integers(65, 70, size=10 + (100 * rank + k) % 13).astype(np.uint8) for k in range(7)]
    ln = np.array([len(x) for x in seqs], np.int32)
    return np.concatenate(seqs), (np.cumsum(ln) - ln).astype(np.int64), ln


def _empty_rank_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from telr_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the locus table: an empty contribution from rank 1 must not disturb the one all-gather
    rows = _make_rows([rank]) if rank != 1 else _make_rows([])
    merged = shard.all_gather_rows(rows, dist, capacity=1)
    if rank == 0:
        np.save(os.path.join(out_dir, "rows.npy"), merged)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_all_gather_with_an_empty_rank(tmp_path, world):
    """a rank that owns no locus contributes an empty block to the ONE all-gather of the locus table (the window-read exchange
    itself is covered in its packed form below; the ASCII form of rounds 1-3 is gone)"""
    import torch.multiprocessing as mp
    mp.spawn(_empty_rank_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rows = np.load(str(tmp_path / "rows.npy"))
    assert rows["locus_id"].tolist() == [r for r in range(world) if r != 1]


def _packed_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from telr_amd import shard
    import packed_np
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    it = _exchange_items(rank, world)
    buf, off, ln = _exchange_reads(rank)
    seqs = [bytes(buf[o:o + n]).decode() for o, n in zip(off, ln)]          # letters A..E: E packs as an ambiguous base
    lens, w2, wn = packed_np.pack(seqs)
    ridx = np.array([x[3] for x in it], np.int64)

    def gather_packed(order):
        a, b = packed_np.subset_words(lens, w2, wn, ridx[order])
        return torch.from_numpy(a.view(np.int32).copy()), torch.from_numpy(b.view(np.int32).copy())
    tm = {}
    loc, rid, rl, g2, gn, order = shard.exchange_window_reads_packed([x[0] for x in it], [x[1] for x in it], [x[2] for x in it], ln[ridx] if len(ridx) else np.zeros(0, np.int32),
                                                                     gather_packed, dist, timings=tm)
    got = packed_np.unpack(rl, g2.numpy(), gn.numpy())
    assert "collective_s" in tm and len(g2) == 2 * len(gn) == int(shard.packed_words(rl)[0].sum())
    import json
    json.dump([(int(loc[i]), int(rid[i]), got[i]) for i in order], open(os.path.join(out_dir, "packed%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_packed_window_read_exchange(tmp_path, world):
    """round 4: the same hand-off on PACKED words (2-bit codes + ambiguity mask, the library's device layout) as torch tensors:
    two collectives (counts, one int32 payload per peer), nothing unpacked on the way; every read arrives at the owner of its
    locus with the bases it left with, also from / to a rank that sends nothing"""
    import json
    import torch.multiprocessing as mp
    mp.spawn(_packed_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = {r: [] for r in range(world)}
    for rank in range(world):
        buf, off, ln = _exchange_reads(rank)
        for locus, gid, d, k in _exchange_items(rank, world):
            want[d].append((locus, gid, bytes(buf[off[k]:off[k] + ln[k]]).decode().replace("B", "N").replace("D", "N").replace("E", "N")))
    for r in range(world):
        got = [tuple(x) for x in json.load(open(str(tmp_path / ("packed%d.json" % r))))]
        assert got == sorted(want[r])


def _stage1_fake(rank, world):
    """this rank's share of a fabricated job: 40 reads dealt round-robin; read g has 0-3 records at pseudo-random coordinates"""
    sys.path.insert(0, ROOT)
    from telr_amd._abi import ALN_DTYPE
    gids = np.arange(rank, 40, world)
    seqs, names, recs, cig = [], [], [], []
    for k, g in enumerate(gids):
        r = np.random.default_rng(1000 + g)
        L = int(r.integers(5, 150))
        seqs.append("".join(r.choice(list("ACGTN"), L))); names.append("read%d" % g)
        for j in range(int(r.integers(0, 4))):
            a = np.zeros(1, ALN_DTYPE)[0]
            a["qid"] = k; a["tid"] = int(r.integers(0, 3)); a["ts"] = int(r.integers(0, 1000)); a["te"] = a["ts"] + 10; a["qlen"] = L
            a["flags"] = 1 if j == 0 else 2; a["n_cigar"] = int(r.integers(1, 6)); a["cigar_off"] = len(cig); a["mapq"] = g
            cig += [int(x) for x in r.integers(1, 1 << 20, size=a["n_cigar"])]
            recs.append(a)
    alns = np.array(recs, ALN_DTYPE) if recs else np.zeros(0, ALN_DTYPE)
    return gids, seqs, names, alns, np.array(cig, np.uint32)


def _stage1_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import json
    import torch
    import torch.distributed as dist
    from telr_amd import shard
    import packed_np
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gids, seqs, names, alns, cig = _stage1_fake(rank, world)
    lens, w2, wn = packed_np.pack(seqs)
    keys = alns["tid"].astype(np.int64) << 32 | alns["ts"].astype(np.int64)
    split = shard.stage1_splitters(keys, world, dist, torch.device("cpu"))
    dest = np.searchsorted(split, keys, side="right")

    def gather_packed(idx):
        a, b = packed_np.subset_words(lens, w2, wn, idx)
        return torch.from_numpy(a.view(np.int32).copy()), torch.from_numpy(b.view(np.int32).copy())
    got = shard.exchange_stage1(alns, torch.from_numpy(cig.view(np.int32).copy()), lens, names, gids, gather_packed, dest, world, dist, torch.device("cpu"))
    rs = packed_np.unpack(got["lengths"], got["seq2"].numpy(), got["nmask"].numpy())
    cg = got["cig"].numpy().view(np.uint32)
    out = {"split": [int(x) for x in split], "gid": [int(x) for x in got["gid"]], "names": got["names"], "reads": rs,
           "recs": [(int(got["gid"][a["qid"]]), int(a["tid"]), int(a["ts"]), int(a["flags"]), int(e), [int(x) for x in cg[a["cigar_off"]:a["cigar_off"] + a["n_cigar"]]]) for a, e in zip(got["alns"], got["emit"])]}
    json.dump(out, open(os.path.join(out_dir, "s1_%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_stage1_range_partition_and_exchange(tmp_path, world):
    """round 4, the N-rank BAM: records are range-partitioned by coordinate with sampled splitters (identical on every rank); every
    record arrives exactly once as "emit" on the rank that owns its slice, with its read (packed words intact) and with ALL the
    other records of that read as non-emit copies; reads without a record go to the last rank; reads arrive in ascending job
    order, records query-major, CIGAR words intact"""
    import json
    import torch.multiprocessing as mp
    mp.spawn(_stage1_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [json.load(open(str(tmp_path / ("s1_%d.json" % r)))) for r in range(world)]
    assert all(o["split"] == outs[0]["split"] for o in outs) and outs[0]["split"] == sorted(outs[0]["split"])
    want_reads, want_recs = {}, {}
    for rank in range(world):
        gids, seqs, names, alns, cig = _stage1_fake(rank, world)
        for g, s_ in zip(gids, seqs):
            want_reads[int(g)] = s_
        for a in alns:
            want_recs.setdefault(int(gids[a["qid"]]), []).append((int(a["tid"]), int(a["ts"]), int(a["flags"]), [int(x) for x in cig[a["cigar_off"]:a["cigar_off"] + a["n_cigar"]]]))
    emitted = {}
    for r, o in enumerate(outs):
        assert o["gid"] == sorted(o["gid"]) and o["names"] == ["read%d" % g for g in o["gid"]]
        assert o["reads"] == [want_reads[g] for g in o["gid"]]
        per = {}
        last = -1
        for g, tid, ts, fl, e, cg in o["recs"]:
            assert g >= last; last = g                                  # query-major in job order
            per.setdefault(g, []).append((tid, ts, fl, cg))
            if e:
                key = (tid << 32) | ts
                assert (r == 0 or key >= o["split"][r - 1]) and (r == world - 1 or key < o["split"][r])      # inside this rank's slice
                emitted.setdefault(g, []).append((tid, ts, fl, cg))
        for g, lst in per.items():
            assert lst == want_recs[g]                                  # every record of the read, in the engine's order
        for g in o["gid"]:                                             # a read is here because a record of it is, or (last rank) it has none
            assert g in per or (r == world - 1 and g not in want_recs)
    assert {g: sorted(v) for g, v in emitted.items()} == {g: sorted(v) for g, v in want_recs.items()}      # every record emitted exactly once
    assert sorted(set(g for o in outs for g in o["gid"]) ) == sorted(want_reads)                            # no read lost (unmapped ones included)


def test_bench_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` without RANK in the environment starts two ranks itself (torch.distributed.run as a child of
    a process that has not touched the GPU) and relays rank 0's line -- here with the gloo backend and no GPU work"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--backend", "gloo"], env=env,
                         capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-1500:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["rank_sum"] == 3 and d["dry_launch"]


def test_locus_names_with_underscores():
    """chrUn_CP007071v1-style chromosome names: the locus of a liftover report is everything before its last two fields"""
    from telr_amd import locus_pipeline
    assert locus_pipeline.locus_of_report({"ID": "chrUn_CP007071v1_100_101_4000_4900"}) == "chrUn_CP007071v1_100_101"
    assert locus_pipeline.locus_of_report({"ID": "chr2L_33000_33020_12_4711"}) == "chr2L_33000_33020"
    assert locus_pipeline.locus_cost({"contig": "ACGT" * 5, "alt": "AC", "read_bases": 100}) == 122


def _stage1_failing_worker(rank, world, port, out_dir):
    """rank 1's packing step raises (a stand-in for `out of device memory` in the gather): every rank must come out of the call
    with JobBamError instead of one unwinding and the others waiting in the all-to-all for ever"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import json
    import torch
    import torch.distributed as dist
    from telr_amd import shard
    import packed_np
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gids, seqs, names, alns, cig = _stage1_fake(rank, world)
    lens, w2, wn = packed_np.pack(seqs)
    keys = alns["tid"].astype(np.int64) << 32 | alns["ts"].astype(np.int64)
    split = shard.stage1_splitters(keys, world, dist, torch.device("cpu"))
    dest = np.searchsorted(split, keys, side="right")

    def gather_packed(idx):
        if rank == 1:
            raise MemoryError("out of device memory (test)")
        a, b = packed_np.subset_words(lens, w2, wn, idx)
        return torch.from_numpy(a.view(np.int32).copy()), torch.from_numpy(b.view(np.int32).copy())
    what = "returned"
    try:
        shard.exchange_stage1(alns, torch.from_numpy(cig.view(np.int32).copy()), lens, names, gids, gather_packed, dest, world, dist, torch.device("cpu"))
    except shard.JobBamError as e:
        what = "JobBamError: " + str(e)
    json.dump({"what": what}, open(os.path.join(out_dir, "fail_%d.json" % rank), "w"))
    dist.barrier()              # every rank is still in step: the next collective works
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_a_failing_rank_takes_every_rank_out_of_the_stage1_exchange(tmp_path):
    import json
    import torch.multiprocessing as mp
    world = 3
    mp.spawn(_stage1_failing_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [json.load(open(str(tmp_path / ("fail_%d.json" % r))))["what"] for r in range(world)]
    assert all(o.startswith("JobBamError") for o in outs), outs
    assert "this rank: MemoryError" in outs[1] and "another rank" in outs[0] and "another rank" in outs[2]
