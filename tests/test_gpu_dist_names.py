"""The unique-name test across several GPUs on ONE GPU: a file is cut into record-aligned
shards ("virtual ranks"), every shard exports (fingerprint, global index) pairs bucketed by owner,
the all-to-all is done by hand, every owner runs its fingerprint set, candidates are confirmed on
the name bytes.  The finding must be the one the oracle's serial loop makes on the whole file.  The
same protocol through fastq_utils_amd.dist.global_first_duplicate on a 1-rank RCCL group."""
import os

import numpy as np
import pytest

from oracle import loader as orc
from tests import fuzz

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


def records_of(img):
    lines = img.split(b"\n")[:-1]
    return [b"\n".join(lines[4 * r:4 * r + 4]) + b"\n" for r in range(len(lines) // 4)]


def shard_image(img, n_shards):
    lines = img.split(b"\n")[:-1]
    n = len(lines) // 4
    out = []
    for first, cnt in fdist.shard_records(n, n_shards):
        out.append((first, cnt, b"\n".join(lines[4 * first:4 * (first + cnt)]) + (b"\n" if cnt else b"")))
    return out


def virtual_first_duplicate(ctx, img, n_shards):
    st = fq.abi.probe_first_record(img, False)
    shards = shard_image(img, n_shards)
    frames, bufs, counts = [], [], []
    for first, cnt, piece in shards:
        if cnt:
            r = ctx.validate(piece, None, st, flags=fq.abi.VALIDATE_NO_STATS)
            assert r["n_records"] == cnt
            frames.append(ctx.retain_frame())
        else:
            frames.append(None)
        buf = torch.empty(max(1, cnt) * fdist.FP_BYTES, dtype=torch.uint8, device="cuda")
        counts.append(ctx.names_fingerprints(frames[-1], st, first, n_shards, buf.data_ptr()) if cnt else [0] * n_shards)
        bufs.append(buf)
    cands = []
    for owner in range(n_shards):
        parts = []
        for r in range(n_shards):
            start = sum(counts[r][:owner]) * fdist.FP_BYTES
            parts.append(bufs[r][start:start + counts[r][owner] * fdist.FP_BYTES])
        recv = torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8, device="cuda")
        n_recv = recv.numel() // fdist.FP_BYTES
        torch.cuda.synchronize()  # torch.cat ran on torch's stream, the library launches on its own
        s = ctx.fingerprint_set(max(1024, n_recv))
        s.insert(recv.data_ptr(), n_recv)
        c, found = s.candidates()
        assert found == len(c)
        cands += c
        s.close()

    def name_of(g):
        for (first, cnt, _), fr in zip(shards, frames):
            if first <= g < first + cnt:
                return ctx.frame_name(fr, st, g - first)
        raise KeyError(g)

    hit = fdist.resolve_candidates(cands, name_of)
    out = None if hit is None else (hit, name_of(hit))
    for fr in frames:
        if fr is not None:
            fr.release()
    return out


def oracle_first_duplicate(img):
    r = orc.fastq_info(img, "x.fastq", flags=orc.FLAG_Q)
    if r["first"]["code"] == 3:
        return r["first"]["record"]
    assert r["first"]["code"] == 0, r["first"]
    return None


@pytest.mark.parametrize("style", ["casava", "slash", "int", "nosuffix"])
@pytest.mark.parametrize("n_shards", [1, 2, 5, 8])
def test_virtual_ranks_find_what_the_serial_loop_finds(ctx, style, n_shards):
    rng = np.random.default_rng(hash((style, n_shards)) & 0xFFFF)
    img = fuzz.make_fastq(rng, 4000, 20, 60, style)
    assert virtual_first_duplicate(ctx, img, n_shards) is None and oracle_first_duplicate(img) is None
    recs = records_of(img)
    # plant repeats: far apart (other shard), near (same shard), and two competing ones
    for plan in ([(3900, 17)], [(1201, 1200)], [(3000, 10), (2500, 2400), (3999, 0)]):
        rr = list(recs)
        for dst, src in plan:
            hdr = rr[src].split(b"\n", 1)[0]
            rr[dst] = hdr + b"\n" + rr[dst].split(b"\n", 1)[1]
        bad = b"".join(rr)
        want = oracle_first_duplicate(bad)
        got = virtual_first_duplicate(ctx, bad, n_shards)
        assert want == min(d for d, _ in plan)
        assert got is not None and got[0] == want


def test_protocol_through_a_one_rank_rccl_group(ctx):
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(12)
        img = fuzz.make_fastq(rng, 3000, 30, 50, "casava")
        recs = records_of(img)
        recs[2222] = recs[5].split(b"\n", 1)[0] + b"\n" + recs[2222].split(b"\n", 1)[1]
        bad = b"".join(recs)
        st = fq.abi.probe_first_record(bad, False)
        # two pieces = two retained frames of the same rank
        cut = len(b"".join(recs[:1500]))
        frames = []
        for piece in (bad[:cut], bad[cut:]):
            r = ctx.validate(piece, None, st, flags=fq.abi.VALIDATE_NO_STATS)
            frames.append((ctx.retain_frame(), r["n_records"]))
        got = fdist.global_first_duplicate(ctx, frames, st, 0)
        assert got is not None and got[0] == 2222 == oracle_first_duplicate(bad)
        for fr, _ in frames:
            fr.release()
    finally:
        dist.destroy_process_group()
