"""Pairing two files across several GPUs on ONE GPU: each file is cut into its own
record-aligned shards ("virtual ranks": the shards of the two files do NOT line up), every shard exports
(fingerprint, index) pairs bucketed by owner, file-2 entries flagged; the all-to-all is done by hand; every
owner sorts and classifies its runs on the device (fqg_fpset_pair_runs); what remains is resolved on the
name bytes.  The outcome must be what the oracle's serial file-2 loop finds on the whole files.  The same
protocol through fastq_utils_amd.dist.global_pairing on a 1-rank RCCL group."""
import os
import re

import numpy as np
import pytest

from oracle import loader as orc
from tests import fuzz
from tests.test_gpu_dist_names import records_of, shard_image

pytestmark = pytest.mark.gpu
fq = pytest.importorskip("fastq_utils_amd")
torch = pytest.importorskip("torch")
from fastq_utils_amd import dist as fdist  # noqa: E402
from tests.util import free_port  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    c = fq.Context(0)
    yield c
    c.close()


def mates(img):
    """file 2 of a pair: same records with the mate digit of the Casava comment changed"""
    return img.replace(b" 1:N:0:", b" 2:N:0:")


def virtual_pairing(ctx, img1, img2, n_shards, named=True):
    """named: the names travel beside the pairs (64-byte records) and the owners hold a holder's name against its
    asker's bytes - what dist.global_pairing and host/fq_names_multi.h do; False: fingerprints alone"""
    st1, st2 = fq.abi.probe_first_record(img1, True), fq.abi.probe_first_record(img2, True)
    held = []  # (flag, first, cnt, frame, state)
    bufs, nbufs, counts = [], [], []
    for img, st, flag, shards in ((img1, st1, 0, n_shards), (img2, st2, fdist.FP_FILE2, max(1, n_shards - 1) if n_shards > 2 else n_shards)):
        for first, cnt, piece in shard_image(img, shards):
            fr = None
            if cnt:
                r = ctx.validate(piece, None, st, flags=fq.abi.VALIDATE_NO_STATS)
                assert r["n_records"] == cnt
                fr = ctx.retain_frame()
            held.append((flag, first, cnt, fr, st))
            buf = torch.empty(max(1, cnt) * fdist.FP_BYTES, dtype=torch.uint8, device="cuda")
            nbuf = torch.empty(max(1, cnt) * fdist.NAME_BYTES, dtype=torch.uint8, device="cuda")
            counts.append(ctx.names_fingerprints(fr, st, first | flag, n_shards, buf.data_ptr(), nbuf.data_ptr() if named else None)
                          if cnt else [0] * n_shards)
            bufs.append(buf)
            nbufs.append(nbuf)

    def name_of(g):
        raw, flag = g & ~fdist.FP_FILE2, g & fdist.FP_FILE2
        for fl, first, cnt, fr, st in held:
            if fl == flag and first <= raw < first + cnt:
                return ctx.frame_name(fr, st, raw - first)
        raise KeyError(g)

    parts = []
    for owner in range(n_shards):
        chunks, nchunks = [], []
        for r in range(len(bufs)):
            start = sum(counts[r][:owner])
            chunks.append(bufs[r][start * fdist.FP_BYTES:(start + counts[r][owner]) * fdist.FP_BYTES])
            nchunks.append(nbufs[r][start * fdist.NAME_BYTES:(start + counts[r][owner]) * fdist.NAME_BYTES])
        recv, nrecv = torch.cat(chunks), torch.cat(nchunks)
        n_recv = recv.numel() // fdist.FP_BYTES
        torch.cuda.synchronize()
        s = ctx.fingerprint_set(max(1024, n_recv))
        s.insert(recv.data_ptr(), n_recv, nrecv.data_ptr() if named else None)
        summary, entries = s.pair_runs()
        assert summary["n_complex"] == len(entries)
        s.close()
        parts.append((summary["matched"], summary["leftover"], summary["unpaired"], summary["first_unpaired"]))
        if entries:
            parts.append(fdist.resolve_pair_runs(entries, name_of))
    out = fdist.merge_pairing(parts)
    for _, _, _, fr, _ in held:
        if fr is not None:
            fr.release()
    return out


def oracle_pairing(img1, img2):
    """(first unpaired file-2 record or None, leftover reported at the end or 0)"""
    r = orc.fastq_info(img1, "a_1.fastq", img2, "a_2.fastq", orc.ARG2_FILE, flags=orc.FLAG_Q)
    if r["first"]["code"] == 13:  # FQG_E_UNPAIRED
        return r["first"]["record"], None
    assert r["first"]["code"] == 0, (r["first"], r["stderr"][-300:])
    m = re.search(r"found (\d+) unpaired reads", r["stderr"])
    return None, int(m.group(1)) if m else 0


@pytest.mark.parametrize("n_shards", [1, 2, 3, 8])
def test_virtual_ranks_pair_like_the_serial_loop(ctx, n_shards):
    rng = np.random.default_rng(n_shards + 40)
    img1 = fuzz.make_fastq(rng, 5000, 20, 60, "casava")
    recs1 = records_of(img1)
    recs2 = records_of(mates(img1))
    # clean pair, file 2 in another order
    perm = rng.permutation(len(recs2))
    f2 = b"".join(recs2[i] for i in perm)
    got = virtual_pairing(ctx, img1, f2, n_shards)
    assert got == (5000, 0, 0, None) and oracle_pairing(img1, f2) == (None, 0)
    # file 1 has reads whose mates are missing: reported at the end
    f2_short = b"".join(recs2[i] for i in perm if i % 17)
    want = oracle_pairing(img1, f2_short)
    got = virtual_pairing(ctx, img1, f2_short, n_shards)
    assert want[0] is None and got[3] is None and got[1] == want[1] > 0
    # file 2 has reads without a mate, and a read that asks twice: the serial loop stops at the first of them
    for plan in ("stranger", "twice", "both"):
        rr = [recs2[i] for i in perm]
        if plan in ("stranger", "both"):
            rr.insert(3333, b"@ZZZ:9:9:9:9:9:9 2:N:0:ACGT\nACGT\n+\nIIII\n")
        if plan in ("twice", "both"):
            rr.insert(1200, rr[100])
        bad2 = b"".join(rr)
        want = oracle_pairing(img1, bad2)
        got = virtual_pairing(ctx, img1, bad2, n_shards)
        assert want[0] is not None and got[3] == want[0], (plan, got, want)
        f1_short = b"".join(r for i, r in enumerate(recs1) if i % 23)  # and mates missing in file 1 as well
        want = oracle_pairing(f1_short, bad2)
        got = virtual_pairing(ctx, f1_short, bad2, n_shards)
        assert got[3] == want[0], (plan, got, want)


def test_protocol_through_a_one_rank_rccl_group(ctx):
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(77)
        img1 = fuzz.make_fastq(rng, 3000, 30, 50, "casava")
        recs2 = records_of(mates(img1))
        rr = [recs2[i] for i in rng.permutation(len(recs2))]
        rr.insert(2000, rr[7])
        del rr[55]
        img2 = b"".join(rr)
        st1, st2 = fq.abi.probe_first_record(img1, True), fq.abi.probe_first_record(img2, True)
        frames = {}
        for key, img, st in ((1, img1, st1), (2, img2, st2)):
            cut = len(b"".join(records_of(img)[:1100]))
            frames[key] = []
            for piece in (img[:cut], img[cut:]):
                r = ctx.validate(piece, None, st, flags=fq.abi.VALIDATE_NO_STATS)
                frames[key].append((ctx.retain_frame(), r["n_records"]))
        got = fdist.global_pairing(ctx, frames[1], st1, 0, frames[2], st2, 0)
        want = oracle_pairing(img1, img2)
        assert got["first_unpaired"] is not None and got["first_unpaired"][0] == want[0] == 1999
        assert got["leftover"] == 1 and got["unpaired"] == 1 and got["matched"] == 2999
        for key in frames:
            for fr, _ in frames[key]:
                fr.release()
    finally:
        dist.destroy_process_group()


def test_mates_in_one_order_are_paired_without_an_exchange(ctx):
    """global_pairing(file1_unique=True): shards cut at the same record numbers, every name at its place -> by position
    (dist.paired_by_position), no all-to-all; one read out of place, a shorter file 2, shards that differ -> the exchange,
    with the oracle's outcome"""
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(78)
        img1 = fuzz.make_fastq(rng, 3000, 30, 50, "casava")
        recs2 = records_of(mates(img1))
        swapped = list(recs2)
        swapped[100], swapped[2000] = swapped[2000], swapped[100]
        for label, second, base2 in (("ordered", recs2, 0), ("swapped", swapped, 0), ("short", recs2[:2500], 0), ("other_base", recs2, 5)):
            img2 = b"".join(second)
            st1, st2 = fq.abi.probe_first_record(img1, True), fq.abi.probe_first_record(img2, True)
            frames = {}
            for key, img, st, at in ((1, img1, st1, 1100), (2, img2, st2, 700)):   # (the two files cut at different records)
                cut = len(b"".join(records_of(img)[:at]))
                frames[key] = []
                for piece in (img[:cut], img[cut:]):
                    r = ctx.validate(piece, None, st, flags=fq.abi.VALIDATE_NO_STATS)
                    frames[key].append((ctx.retain_frame(), r["n_records"]))
            got = fdist.global_pairing(ctx, frames[1], st1, 0, frames[2], st2, base2, file1_unique=True)
            plain = fdist.global_pairing(ctx, frames[1], st1, 0, frames[2], st2, base2)
            assert bool(got.get("by_position")) == (label == "ordered"), (label, got)
            assert not plain.get("by_position")
            for k in ("matched", "leftover", "unpaired", "first_unpaired"):
                assert got[k] == plain[k], (label, k, got, plain)
            if base2 == 0:
                want = oracle_pairing(img1, img2)
                assert (got["first_unpaired"][0] if got["first_unpaired"] else None) == want[0], (label, got, want)
            for key in frames:
                for fr, _ in frames[key]:
                    fr.release()
    finally:
        dist.destroy_process_group()


def orphans_on_both_sides(seed, n=3000, long_names=False):
    """two files of which every seventh read of file 1 and every eleventh of file 2 has no mate, file 2 in another order
    (long_names: names of 60 - 90 bytes - beyond the 56 a name record holds)"""
    rng = np.random.default_rng(seed)
    img1 = fuzz.make_fastq(rng, n, 20, 60, "casava")
    if long_names:
        lines = img1.split(b"\n")
        for k in range(0, len(lines) - 1, 4):  # (header lines only)
            name, rest = lines[k].split(b" ", 1)
            lines[k] = name + b":" + b"x" * (40 + len(name) % 30) + b" " + rest
        img1 = b"\n".join(lines)
    recs1, recs2 = records_of(img1), records_of(mates(img1))
    perm = rng.permutation(n)
    f1 = b"".join(r for i, r in enumerate(recs1) if i % 7)
    f2 = b"".join(recs2[i] for i in perm if i % 11)
    return f1, f2


@pytest.mark.parametrize("named", [True, False], ids=["names_travel", "fingerprints_alone"])
@pytest.mark.parametrize("n_shards", [1, 3])
def test_names_beside_the_pairs_change_nothing_where_fingerprints_are_right(ctx, n_shards, named):
    for long_names in (False, True):
        f1, f2 = orphans_on_both_sides(5 + n_shards, long_names=long_names)
        want = oracle_pairing(f1, f2)
        got = virtual_pairing(ctx, f1, f2, n_shards, named=named)
        assert got[3] == want[0] is not None, (got, want)


def test_names_decide_where_fingerprints_collide():
    """FQGPU_FP_WEAK_BITS=10 (a process of its own: the library reads it once): fingerprints of ten bits and no check
    bits - with hundreds of reads without a mate on both sides, runs of one holder and one asker with DIFFERENT names
    are the rule.  With the names beside the pairs the outcome is the serial loop's; without them it is not (which
    is what says that the names decided)."""
    import subprocess
    import sys

    from tests.util import REPO

    p = subprocess.run([sys.executable, "-m", "tests.weak_fp_pairing"], cwd=REPO, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, FQGPU_FP_WEAK_BITS="10", FQGPU_FP_WEAK_UNNAMED="1"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "names_travel: as the serial loop" in p.stdout and "fingerprints_alone: NOT as the serial loop" in p.stdout, p.stdout
