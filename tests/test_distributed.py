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


def _exchange_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from telr_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    it = _exchange_items(rank, world)
    loc, rid, (buf, off, ln) = shard.exchange_window_reads([x[0] for x in it], [x[1] for x in it], [x[2] for x in it], _exchange_reads(rank), [x[3] for x in it], dist)
    np.save(os.path.join(out_dir, "got%d.npy" % rank), np.array([(l, r, n, int(buf[o:o + n].sum())) for l, r, o, n in zip(loc, rid, off, ln)], np.int64).reshape(-1, 4))
    # the locus table: an empty contribution from rank 1 must not disturb the one all-gather
    rows = _make_rows([rank]) if rank != 1 else _make_rows([])
    merged = shard.all_gather_rows(rows, dist, capacity=1)
    if rank == 0:
        np.save(os.path.join(out_dir, "rows.npy"), merged)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_window_read_exchange_and_empty_rank(tmp_path, world):
    """the all-to-all of window reads (stage 1 -> loci hand-off when reads are sharded) delivers every read to the owner of
    its locus, sorted by (locus, read id), also with a rank that sends nothing; world size 1 is the identity"""
    import torch.multiprocessing as mp
    from telr_amd import shard
    mp.spawn(_exchange_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = {r: [] for r in range(world)}
    for rank in range(world):
        buf, off, ln = _exchange_reads(rank)
        for locus, gid, d, k in _exchange_items(rank, world):
            want[d].append((locus, gid, int(ln[k]), int(buf[off[k]:off[k] + ln[k]].sum())))
    for r in range(world):
        got = np.load(str(tmp_path / ("got%d.npy" % r)))
        assert [tuple(x) for x in got.tolist()] == sorted(want[r])
    rows = np.load(str(tmp_path / "rows.npy"))
    assert rows["locus_id"].tolist() == [r for r in range(world) if r != 1]
    reads = (np.arange(6, dtype=np.uint8), np.array([0, 4], np.int64), np.array([4, 2], np.int32))
    loc, rid, (buf, off, ln) = shard.exchange_window_reads([3, 1], [9, 2], [0, 0], reads, [0, 1])
    assert loc.tolist() == [1, 3] and rid.tolist() == [2, 9] and [buf[o:o + n].tolist() for o, n in zip(off, ln)] == [[4, 5], [0, 1, 2, 3]]


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


def _stage1_piece(rank, world):
    """what rank `rank` mapped: records with LOCAL query ids and CIGAR offsets, its reads and their names (rank 1 of 3 holds nothing)"""
    from telr_amd._abi import ALN_DTYPE
    rng = np.random.default_rng(50 + rank)
    n_reads = 0 if (world == 3 and rank == 1) else 4 + rank
    ln = rng.integers(5, 40, size=n_reads).astype(np.int32)
    buf = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(ln.sum())) if n_reads else np.zeros(0, np.uint8)
    off = np.cumsum(ln.astype(np.int64)) - ln
    names = ["r%d_%d" % (rank, i) for i in range(n_reads)]
    recs, cig = [], []
    for q in range(n_reads):
        for _ in range(int(rng.integers(0, 3))):
            a = np.zeros(1, ALN_DTYPE)
            a["qid"] = q; a["tid"] = 0; a["qlen"] = ln[q]; a["ts"] = int(rng.integers(0, 1000)); a["flags"] = 1
            ops = [int(rng.integers(1, 9)) << 4 | int(rng.integers(0, 3)) for _ in range(int(rng.integers(1, 5)))]
            a["cigar_off"] = len(cig); a["n_cigar"] = len(ops); cig += ops
            recs.append(a)
    alns = np.concatenate(recs) if recs else np.zeros(0, ALN_DTYPE)
    return alns, np.array(cig, np.uint32), (buf, off, ln), names


def _gather_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from telr_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    alns, cig, reads, names = _stage1_piece(rank, world)
    got = shard.gather_stage1(alns, cig, reads, names, dist=dist)
    # the same with file-order numbers: read q of rank r is read q * world + r of the job
    got2 = shard.gather_stage1(alns, cig, reads, names, dist=dist, read_gid=np.arange(len(names), dtype=np.int64) * world + rank)
    if rank == 0:
        a, c, (buf, off, ln), nm = got
        a2, c2, (buf2, off2, ln2), nm2 = got2
        np.savez(out_path, alns=a, cig=c, buf=buf, off=off, ln=ln, names=np.array(nm), alns2=a2, cig2=c2, buf2=buf2, off2=off2, ln2=ln2, names2=np.array(nm2))
    else:
        assert got is None and got2 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 3])
def test_gather_stage1_rebases_query_ids_and_cigar_offsets(tmp_path, world):
    """the job's records on rank 0 = the ranks' records in rank order, query ids and CIGAR offsets re-based, reads and names
    concatenated (world 3: the middle rank holds no reads at all)"""
    import torch.multiprocessing as mp
    out = str(tmp_path / "g.npz")
    mp.spawn(_gather_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    z = np.load(out)
    q0 = 0; c0 = 0; k0 = 0; b0 = 0
    for r in range(world):
        alns, cig, (buf, off, ln), names = _stage1_piece(r, world)
        n = len(alns)
        got = z["alns"][k0:k0 + n]
        np.testing.assert_array_equal(got["qid"], alns["qid"] + q0)
        np.testing.assert_array_equal(got["cigar_off"], alns["cigar_off"] + c0)
        for f in ("ts", "n_cigar", "qlen", "flags"):
            np.testing.assert_array_equal(got[f], alns[f])
        np.testing.assert_array_equal(z["cig"][c0:c0 + len(cig)], cig)
        np.testing.assert_array_equal(z["ln"][q0:q0 + len(ln)], ln)
        np.testing.assert_array_equal(z["buf"][b0:b0 + len(buf)], buf)
        assert list(z["names"][q0:q0 + len(ln)]) == names
        q0 += len(ln); c0 += len(cig); k0 += n; b0 += len(buf)
    assert k0 == len(z["alns"]) and q0 == len(z["ln"]) and (z["off"] == np.cumsum(z["ln"].astype(np.int64)) - z["ln"]).all()
    # every record's CIGAR is where its offset says
    for a in z["alns"]:
        assert a["cigar_off"] + a["n_cigar"] <= len(z["cig"])
    # with read_gid: reads in file order, the records of a read together and in the engine's order, CIGARs still where the offsets say
    pieces = [_stage1_piece(r, world) for r in range(world)]
    gids = sorted((q * world + r, r, q) for r in range(world) for q in range(len(pieces[r][3])))
    assert list(z["names2"]) == [pieces[r][3][q] for _, r, q in gids]
    np.testing.assert_array_equal(z["ln2"], [pieces[r][2][2][q] for _, r, q in gids])
    # (the bases are not moved: the offsets are permuted with the lengths)
    for k, (_, r, q) in enumerate(gids):
        want = pieces[r][2][0][pieces[r][2][1][q]:pieces[r][2][1][q] + pieces[r][2][2][q]]
        np.testing.assert_array_equal(z["buf2"][z["off2"][k]:z["off2"][k] + z["ln2"][k]], want)
    k = 0
    for new_q, (_, r, q) in enumerate(gids):
        alns, cig = pieces[r][0], pieces[r][1]
        for a in alns[alns["qid"] == q]:
            g = z["alns2"][k]; k += 1
            assert g["qid"] == new_q and g["ts"] == a["ts"] and g["n_cigar"] == a["n_cigar"]
            np.testing.assert_array_equal(z["cig2"][g["cigar_off"]:g["cigar_off"] + g["n_cigar"]], cig[a["cigar_off"]:a["cigar_off"] + a["n_cigar"]])
    assert k == len(z["alns2"]) == len(z["alns"])
