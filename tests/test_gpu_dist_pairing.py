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


def virtual_pairing(ctx, img1, img2, n_shards):
    st1, st2 = fq.abi.probe_first_record(img1, True), fq.abi.probe_first_record(img2, True)
    held = []  # (flag, first, cnt, frame, state)
    bufs, counts = [], []
    for img, st, flag, shards in ((img1, st1, 0, n_shards), (img2, st2, fdist.FP_FILE2, max(1, n_shards - 1) if n_shards > 2 else n_shards)):
        for first, cnt, piece in shard_image(img, shards):
            fr = None
            if cnt:
                r = ctx.validate(piece, None, st, flags=fq.abi.VALIDATE_NO_STATS)
                assert r["n_records"] == cnt
                fr = ctx.retain_frame()
            held.append((flag, first, cnt, fr, st))
            buf = torch.empty(max(1, cnt) * fdist.FP_BYTES, dtype=torch.uint8, device="cuda")
            counts.append(ctx.names_fingerprints(fr, st, first | flag, n_shards, buf.data_ptr()) if cnt else [0] * n_shards)
            bufs.append(buf)

    def name_of(g):
        raw, flag = g & ~fdist.FP_FILE2, g & fdist.FP_FILE2
        for fl, first, cnt, fr, st in held:
            if fl == flag and first <= raw < first + cnt:
                return ctx.frame_name(fr, st, raw - first)
        raise KeyError(g)

    parts = []
    for owner in range(n_shards):
        chunks = []
        for r in range(len(bufs)):
            start = sum(counts[r][:owner]) * fdist.FP_BYTES
            chunks.append(bufs[r][start:start + counts[r][owner] * fdist.FP_BYTES])
        recv = torch.cat(chunks)
        n_recv = recv.numel() // fdist.FP_BYTES
        torch.cuda.synchronize()
        s = ctx.fingerprint_set(max(1024, n_recv))
        s.insert(recv.data_ptr(), n_recv)
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
