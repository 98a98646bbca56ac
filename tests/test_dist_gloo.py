"""The N>1 path on CPU: two gloo ranks shard a record range, exchange statistics blobs with
all_gather_object and agree on the merged summary and on the first finding in file order."""
import os
import socket
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from fastq_utils_amd import dist as fdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = fdist.shard_records(1001, world)[rank]
    # synthetic per-rank statistics: record i has read length 100 + i % 7 (+1 for the newline)
    hist = {}
    for i in range(first, first + count):
        rl = 101 + i % 7
        hist[rl] = hist.get(rl, 0) + 1
    blob = fdist.make_acc_blob(count, min(hist), max(hist), 35 + rank, 70 + rank, hist)
    finding = (first + 5, 3, 11) if rank == 1 else None  # rank 1 saw a QLEN error at its 6th record
    blobs, finds = [None] * world, [None] * world
    dist.all_gather_object(blobs, blob)
    dist.all_gather_object(finds, finding)
    merged = fdist.merge_acc_blobs(blobs)
    q.put((rank, merged["num_rds"], merged["min_rl"], merged["max_rl"], merged["min_qbyte"], merged["max_qbyte"],
           fdist.median_rl(merged), fdist.first_finding(finds)))
    dist.destroy_process_group()


def test_two_ranks_merge_statistics_and_findings():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    shards = fdist.shard_records(1001, world)
    assert shards == [(0, 501), (501, 500)]
    lens = sorted(101 + i % 7 for i in range(1001))
    for rank, num, mn, mx, qmn, qmx, med, first in out:
        assert (num, mn, mx, qmn, qmx) == (1001, 101, 107, 35, 71)
        # median_rl: first length whose cumulative count exceeds n/2
        assert med == lens[1001 // 2]
        assert first == (506, 3, 11)


def test_blob_roundtrip_matches_library_layout():
    blob = fdist.make_acc_blob(10, 50, 151, 35, 73, {151: 9, 50: 1})
    p = fdist.parse_acc_blob(blob)
    assert p["num_rds"] == 10 and p["hist"] == {50: 1, 151: 9}
    assert len(blob) == 32 + 8 + 4 * 8


def _fp_worker(rank, world, port, q):
    import torch

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank r sends (10 * r + o + 1) pairs to owner o; pair k of that bucket = (fp, idx) = (r * 1000 + o, k)
    counts = [10 * rank + o + 1 for o in range(world)]
    vals = []
    for o, c in enumerate(counts):
        for k in range(c):
            vals += [rank * 1000 + o, k]
    send = torch.tensor(vals, dtype=torch.int64).view(torch.uint8)
    recv, recv_counts = fdist.exchange_fingerprints(send, counts)
    got = recv.view(torch.int64).view(-1, 2).tolist()
    # the same exchange in rounds of 4 pairs per peer must deliver the same bytes
    recv2, recv_counts2 = fdist.exchange_fingerprints(send, counts, round_pairs=4)
    assert recv_counts2 == recv_counts and recv2.view(torch.int64).view(-1, 2).tolist() == got
    # a collective that delivers wrong bytes (what one 1.6 GB all_to_all_single did on RCCL 2.26) must be an error on the
    # rank that received them, not a wrong finding: flip one byte of what rank 1 receives in the payload exchange
    real = dist.all_to_all_single
    calls = {"n": 0}

    def flipping(out, inp, *a, **kw):
        r = real(out, inp, *a, **kw)
        calls["n"] += 1
        if rank == 1 and out.dtype == torch.uint8 and out.numel() and not calls.get("done"):  # the payload (counts and checksums are int64)
            out[out.numel() // 2] ^= 0x40
            calls["done"] = True
        return r

    dist.all_to_all_single = flipping
    caught = None
    try:
        fdist.exchange_fingerprints(send, counts)
    except RuntimeError as e:
        caught = str(e)
    finally:
        dist.all_to_all_single = real
    assert (caught is not None and "checksum" in caught) == (rank == 1), (rank, caught)
    q.put((rank, recv_counts, got))
    dist.destroy_process_group()


def test_fingerprint_exchange_is_one_all_to_all_of_buckets():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = {r: (c, g) for r, c, g in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for owner in range(world):
        counts, got = out[owner]
        assert counts == [10 * r + owner + 1 for r in range(world)]
        want = [[r * 1000 + owner, k] for r in range(world) for k in range(10 * r + owner + 1)]
        assert got == want


def test_candidates_are_confirmed_on_names_and_the_earliest_repeat_wins():
    names = {5: b"a", 9: b"a", 3: b"x", 7: b"y", 2: b"q", 11: b"q", 20: b"z", 21: b"z"}
    # (3, 7): equal fingerprints, different names - a collision, not a duplicate
    assert fdist.resolve_candidates([(3, 7), (5, 9), (2, 11), (20, 21)], names.__getitem__) == 9
    assert fdist.resolve_candidates([(3, 7)], names.__getitem__) is None
    # a collision (30 vs 31) must not hide the repeat of the second name (31 == 33)
    coll = {30: b"u", 31: b"v", 33: b"v"}
    assert fdist.resolve_candidates([(30, 31), (30, 33)], coll.__getitem__) == 33
    assert fdist.resolve_candidates([], names.__getitem__) is None


def test_umi_shard_merge_gives_first_appearance_ids_over_the_ranks():
    infos = [
        {"code": 0, "features": [b"G3", b"G1"], "cells": [11, 12]},
        {"code": 0, "features": [b"G1", b"G7", b"G3"], "cells": [13]},
        {"code": 0, "features": [], "cells": []},
        {"code": 0, "features": [b"G9"], "cells": [14, 15]},
    ]
    m = fdist.merge_umi_shards(infos)
    assert m["finding"] is None
    assert m["features"] == [b"G3", b"G1", b"G7", b"G9"]
    assert m["remap"] == [[0, 1, 2], [0, 2, 3, 1], [0], [0, 4]]
    assert m["cell_offset"] == [0, 2, 3, 3] and m["cells"] == [11, 12, 13, 14, 15]
    # a cell of an earlier shard again = "not sorted by CR"; a local finding wins by rank order
    infos[3]["cells"] = [14, 12]
    assert fdist.merge_umi_shards(infos)["finding"] == (3, 17, None, None)
    infos[1].update(code=18, record=5, aux=30)
    assert fdist.merge_umi_shards(infos)["finding"] == (1, 18, 5, 30)
    assert fdist.unit_float(5) == 5.0 and fdist.unit_float(1 << 30) == float(1 << 24)


def test_the_files_umi_numbers_from_the_shards_lists():
    """umi_global_table: ids in order of first appearance over the ranks in order = over the file (blabel2id,
    src/bam_umi_count.c:225-260); the table is sorted by packed UMI for the device's look-up"""
    import numpy as np

    from fastq_utils_amd import dist as fdist

    shard0 = np.array([50, 7, 900, 3], dtype=np.uint64)        # first appearance in shard 0: ids 1..4
    shard1 = np.array([7, 11, 50, 2], dtype=np.uint64)         # 11 -> 5, 2 -> 6 (7 and 50 are known)
    shard2 = np.array([], dtype=np.uint64)
    shard3 = np.array([2, 1000, 3], dtype=np.uint64)           # 1000 -> 7
    keys, ids = fdist.umi_global_table([shard0, shard1, shard2, shard3])
    assert list(keys) == sorted({50, 7, 900, 3, 11, 2, 1000})
    assert dict(zip(keys.tolist(), ids.tolist())) == {50: 1, 7: 2, 900: 3, 3: 4, 11: 5, 2: 6, 1000: 7}
    k0, i0 = fdist.umi_global_table([])
    assert len(k0) == 0 and len(i0) == 0


def _exchange_name_records(rank, world, port, q):
    """the same exchange with 64-byte records (the names that travel beside the pairs in a pairing), in rounds"""
    import torch
    import torch.distributed as dist

    from fastq_utils_amd import dist as fdist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        counts = [5 + 3 * rank + o for o in range(world)]
        send = torch.randint(0, 256, (sum(counts) * fdist.NAME_BYTES,), dtype=torch.uint8, generator=g)
        whole, c1 = fdist.exchange_fingerprints(send, counts, record_bytes=fdist.NAME_BYTES)
        rounds, c2 = fdist.exchange_fingerprints(send, counts, round_pairs=2, record_bytes=fdist.NAME_BYTES)
        assert c1 == c2 == [5 + 3 * r + rank for r in range(world)]
        assert torch.equal(whole, rounds) and whole.numel() == sum(c1) * fdist.NAME_BYTES
        # what rank r sent to me is the slice of ITS buffer for owner `rank`
        p = 0
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            cr = [5 + 3 * r + o for o in range(world)]
            theirs = torch.randint(0, 256, (sum(cr) * fdist.NAME_BYTES,), dtype=torch.uint8, generator=gr)
            start = sum(cr[:rank]) * fdist.NAME_BYTES
            want = theirs[start:start + cr[rank] * fdist.NAME_BYTES]
            assert torch.equal(whole[p:p + want.numel()], want), (rank, r)
            p += want.numel()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_exchange_of_name_records_two_ranks():
    import torch.multiprocessing as mp

    from tests.util import free_port

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    ps = [ctx.Process(target=_exchange_name_records, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(30)
    assert got == [(0, "ok"), (1, "ok")], got
