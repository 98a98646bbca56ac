"""Multi-GPU plumbing for the record-parallel paths (one process per GPU).

Validation shards trivially: every rank frames and validates its own record-aligned shard; no
data-path collective is needed.  What has to be combined at the end is tiny: per-rank statistics
blobs (fqg_acc_export) and the first finding in file order.

The unique-name test of the index mode is the one step with a real exchange:
`exchange_fingerprints` is ONE all-to-all (RCCL over xGMI with the nccl backend) of 16-byte
(fingerprint, global record index) pairs bucketed by owner rank, `global_first_duplicate` runs the
whole protocol.  The helpers take the process group as an argument so that the exchange itself can
be exercised with the gloo backend on CPU tensors.
"""
import struct

ACC_STATE = struct.Struct("<QQQII")  # AccState: num_rds, min_rl, max_rl, min_qbyte, max_qbyte


def shard_records(n_records, world):
    """Contiguous, nearly equal record ranges [(first, count)] for `world` ranks."""
    base, extra = divmod(n_records, world)
    out, first = [], 0
    for r in range(world):
        cnt = base + (1 if r < extra else 0)
        out.append((first, cnt))
        first += cnt
    return out


def parse_acc_blob(blob):
    num, mn, mx, qmn, qmx = ACC_STATE.unpack_from(blob, 0)
    (n,) = struct.unpack_from("<Q", blob, ACC_STATE.size)
    pairs = struct.unpack_from("<%dQ" % (2 * n), blob, ACC_STATE.size + 8) if n else ()
    hist = {pairs[2 * i]: pairs[2 * i + 1] for i in range(n)}
    return {"num_rds": num, "min_rl": mn, "max_rl": mx, "min_qbyte": qmn, "max_qbyte": qmx, "hist": hist}


def make_acc_blob(num_rds, min_rl, max_rl, min_qbyte, max_qbyte, hist):
    items = sorted(hist.items())
    flat = [x for kv in items for x in kv]
    return ACC_STATE.pack(num_rds, min_rl, max_rl, min_qbyte, max_qbyte) + struct.pack("<Q", len(items)) + (
        struct.pack("<%dQ" % len(flat), *flat) if flat else b"")


def merge_acc_blobs(blobs):
    """Element-wise merge of exported accumulators (sum / min / max / histogram sum)."""
    parts = [parse_acc_blob(b) for b in blobs]
    hist = {}
    for p in parts:
        for k, v in p["hist"].items():
            hist[k] = hist.get(k, 0) + v
    return {
        "num_rds": sum(p["num_rds"] for p in parts),
        "min_rl": min(p["min_rl"] for p in parts),
        "max_rl": max(p["max_rl"] for p in parts),
        "min_qbyte": min(p["min_qbyte"] for p in parts),
        "max_qbyte": max(p["max_qbyte"] for p in parts),
        "hist": hist,
    }


def first_finding(findings):
    """findings: per-rank (global_record, stage, code) or None; the serial loop reports the
    smallest (record, stage)."""
    live = [f for f in findings if f is not None]
    return min(live) if live else None


def median_rl(merged, second=None):
    """median_rl() of the reference (src/fastq_info.c:39-55) over merged statistics."""
    num1 = merged["num_rds"]
    if num1 == 1 and second is None:
        return merged["min_rl"]
    nreads = num1 + (second["num_rds"] if second else 0)
    ctr, crl = 0, 1
    while crl < 2500000:
        ctr += merged["hist"].get(crl, 0) + (second["hist"].get(crl, 0) if second else 0)
        if num1 > 1 and ctr > nreads // 2:
            break
        crl += 1
    return crl


# ---- read names across ranks ---------------------------------------------------------------------
FP_BYTES = 16  # fqg_fp: u64 fingerprint, u64 global record index


ROUND_PAIRS = 1 << 24  # pairs per (sender, owner) and round: 256 MiB messages


NAME_BYTES = 64  # FQG_NAME_REC_BYTES: a name record that travels beside its pair (pairing: the names decide)


def _slice_checksums(buf, counts, record_bytes=FP_BYTES):
    """[len(counts), 2] int64: for every slice of `buf` (counts[i] records of record_bytes bytes, back to back) the sum
    of its 64-bit words and the sum of their bit-mixed images, both modulo 2^64.  Independent of the order of the
    records inside a slice - which the protocol does not depend on either."""
    import torch

    words = buf.view(torch.int64)
    out = torch.zeros((len(counts), 2), dtype=torch.int64, device=buf.device)
    p, wpr = 0, record_bytes // 8
    for i, c in enumerate(counts):
        w = words[p:p + wpr * c]
        p += wpr * c
        if c:
            out[i, 0] = w.sum()
            out[i, 1] = (w ^ (w >> 29) ^ (w << 17)).sum()
    return out


def exchange_fingerprints(send, send_counts, group=None, round_pairs=None, record_bytes=FP_BYTES):
    """The all-to-all of fingerprint buckets.  `send`: uint8 tensor holding this rank's buckets back to
    back (bucket o = send_counts[o] pairs of FP_BYTES bytes, for owner o); returns (received uint8
    tensor with the pairs grouped by sender, counts received from every rank).  Works on device
    tensors with nccl (= RCCL) and on CPU tensors with gloo.  Buckets larger than `round_pairs` go in
    several rounds of at most 256 MiB per peer (a single 1.6 GB all_to_all_single was observed to
    deliver wrong data with RCCL 2.26 / torch 2.10).  Every round carries a checksum per slice, verified on the
    receiving side: wrong bytes raise instead of becoming wrong findings."""
    import torch
    import torch.distributed as dist

    rp = round_pairs or max(1, ROUND_PAIRS * FP_BYTES // record_bytes)  # (rounds of at most 256 MiB per peer)
    world = dist.get_world_size(group)
    assert len(send_counts) == world
    cnt_in = torch.tensor(send_counts, dtype=torch.int64, device=send.device)
    cnt_out = torch.empty(world, dtype=torch.int64, device=send.device)
    dist.all_to_all_single(cnt_out, cnt_in, group=group)
    recv_counts = [int(x) for x in cnt_out.tolist()]
    biggest = torch.tensor([max(send_counts + [0])], dtype=torch.int64, device=send.device)
    dist.all_reduce(biggest, op=dist.ReduceOp.MAX, group=group)
    rounds = max(1, -(-int(biggest.item()) // rp))
    recv = torch.empty(max(1, sum(recv_counts)) * record_bytes, dtype=torch.uint8, device=send.device)
    send_start = [sum(send_counts[:o]) for o in range(world)]
    recv_start = [sum(recv_counts[:r]) for r in range(world)]
    for k in range(rounds):
        s_lo = [min(c, k * rp) for c in send_counts]
        s_n = [min(c, (k + 1) * rp) - lo for c, lo in zip(send_counts, s_lo)]
        r_lo = [min(c, k * rp) for c in recv_counts]
        r_n = [min(c, (k + 1) * rp) - lo for c, lo in zip(recv_counts, r_lo)]
        if rounds == 1:
            src, dst = send[: sum(send_counts) * record_bytes], recv[: sum(recv_counts) * record_bytes]
        else:
            parts = [send[(send_start[o] + s_lo[o]) * record_bytes:(send_start[o] + s_lo[o] + s_n[o]) * record_bytes]
                     for o in range(world)]
            src = torch.cat(parts) if sum(s_n) else send[:0]
            dst = torch.empty(sum(r_n) * record_bytes, dtype=torch.uint8, device=send.device)
        dist.all_to_all_single(dst, src, output_split_sizes=[c * record_bytes for c in r_n],
                               input_split_sizes=[c * record_bytes for c in s_n], group=group)
        # what arrived is what was sent: a checksum per (sender, owner) slice of the round travels beside it and is
        # held against the received bytes before anybody works on them (the wrong bytes of the 1.6 GB message above
        # would have been findings, or missed findings, of the protocol)
        sums_out = _slice_checksums(src, s_n, record_bytes)
        sums_in = torch.empty_like(sums_out)
        dist.all_to_all_single(sums_in, sums_out, group=group)
        if not torch.equal(sums_in, _slice_checksums(dst, r_n, record_bytes)):
            bad = (sums_in != _slice_checksums(dst, r_n, record_bytes)).any(dim=1).nonzero().flatten().tolist()
            raise RuntimeError(f"fingerprint exchange: round {k}: the bytes received from rank(s) {bad} are not the bytes "
                               f"they sent (checksum mismatch) - the collective delivered wrong data")
        if rounds > 1:
            p = 0
            for r in range(world):
                recv[(recv_start[r] + r_lo[r]) * record_bytes:(recv_start[r] + r_lo[r] + r_n[r]) * record_bytes] = \
                    dst[p * record_bytes:(p + r_n[r]) * record_bytes]
                p += r_n[r]
    recv = recv[: sum(recv_counts) * record_bytes]
    if recv.is_cuda:
        torch.cuda.synchronize(recv.device)  # the library launches on its own stream: the data must have landed
    return recv, recv_counts


def resolve_candidates(candidates, name_of):
    """candidates: (earliest holder, later holder) global record indices with equal fingerprints;
    name_of(idx) -> bytes.  Returns the smallest index whose name really equals the name of an earlier
    holder of the same fingerprint, or None: the record at which the serial loop over the concatenated
    shards reports "duplicated sequence".  All holders of one fingerprint are compared with each
    other, so a collision of two different names cannot hide a repeat of the second one."""
    groups = {}
    for first, later in candidates:
        groups.setdefault(first, set()).add(later)
    best = None
    for first, laters in groups.items():
        seen = {name_of(first)}
        for g in sorted(laters):
            if best is not None and g >= best:
                break
            nm = name_of(g)
            if nm in seen:
                best = g
                break
            seen.add(nm)
    return best


def global_first_duplicate(ctx, frames, state, record_base, group=None, device=None):
    """Every rank calls this with its retained frames [(frame, n_records)] of ONE file (in file order;
    the frames' records are record_base, record_base + 1, ... globally).  Returns (global index of the
    first record whose name already occurred at a smaller global index, its name), or None - the same
    on every rank."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    n_local = sum(n for _, n in frames)
    send = torch.empty(max(1, n_local) * FP_BYTES, dtype=torch.uint8, device=dev)
    # one bucketed export per frame, then regroup by owner (frames are few: one per piece of the file)
    per_frame, base = [], record_base
    off = 0
    for fr, n in frames:
        counts = ctx.names_fingerprints(fr, state, base, world, send.data_ptr() + off * FP_BYTES)
        per_frame.append((off, counts))
        off += sum(counts)
        base += n
    if len(per_frame) > 1:
        parts = [[] for _ in range(world)]
        for o0, counts in per_frame:
            p = o0
            for o, c in enumerate(counts):
                parts[o].append(send[p * FP_BYTES:(p + c) * FP_BYTES])
                p += c
        send_counts = [sum(c[o] for _, c in per_frame) for o in range(world)]
        send = torch.cat([t for o in range(world) for t in parts[o]]) if off else send
    else:
        send_counts = per_frame[0][1] if per_frame else [0] * world
    recv, recv_counts = exchange_fingerprints(send, send_counts, group)
    n_recv = sum(recv_counts)
    fps = ctx.fingerprint_set(max(1024, n_recv))
    try:
        fps.insert(recv.data_ptr(), n_recv)
        cand, found = fps.candidates()
    finally:
        fps.close()
    if found > len(cand):
        raise RuntimeError(f"{found} candidate duplicates exceed the buffer: the input repeats names massively")
    all_cand = [None] * world
    dist.all_gather_object(all_cand, cand, group=group)
    flat = [p for c in all_cand for p in c]
    if not flat:
        return None
    # every rank names the candidates it holds; then everybody can resolve
    mine = {}
    bases, b = [], record_base
    for fr, n in frames:
        bases.append((b, n, fr))
        b += n
    for pair in flat:
        for g in pair:
            for b0, n, fr in bases:
                if b0 <= g < b0 + n and g not in mine:
                    mine[g] = ctx.frame_name(fr, state, g - b0)
    names = [None] * world
    dist.all_gather_object(names, mine, group=group)
    table = {}
    for d in names:
        table.update(d)
    hit = resolve_candidates(flat, table.__getitem__)
    return None if hit is None else (hit, table[hit])


# ---- pairing across ranks (the file-2 loop of fastq_info) --------------------------------------------
FP_FILE2 = 1 << 63


def resolve_pair_runs(entries, name_of):
    """Exact outcome of the runs the device could not classify (fqg_fpset_pair_runs): entries =
    [(run id, index)], file-2 indices carry FP_FILE2; name_of(index with flag) -> bytes.  Per NAME the serial
    loop pairs the file-1 holder with the smallest file-2 asker; later askers of that name, and askers of a
    name without holder, are unpaired; holders nobody asked for are left over.
    Returns (matched, leftover, unpaired, smallest unpaired file-2 index or None)."""
    runs = {}
    for run, idx in entries:
        runs.setdefault(run, []).append(idx)
    matched = leftover = unpaired = 0
    first = None
    for idxs in runs.values():
        by_name = {}
        for g in idxs:
            h, a = by_name.setdefault(name_of(g), ([], []))
            (a if g & FP_FILE2 else h).append(g & ~FP_FILE2)
        for holders, askers in by_name.values():
            askers.sort()
            if holders:  # names of file 1 are unique (its index pass has checked that): one holder
                if askers:
                    matched += 1
                    bad = askers[1:]
                    leftover += len(holders) - 1
                else:
                    bad = []
                    leftover += len(holders)
            else:
                bad = askers
            unpaired += len(bad)
            if bad and (first is None or bad[0] < first):
                first = bad[0]
    return matched, leftover, unpaired, first


def merge_pairing(parts):
    """parts: per owner (matched, leftover, unpaired, first unpaired or None) -> the whole job's"""
    firsts = [p[3] for p in parts if p[3] is not None]
    return (sum(p[0] for p in parts), sum(p[1] for p in parts), sum(p[2] for p in parts), min(firsts) if firsts else None)


def _export_fingerprints(ctx, frames, state, base, world, dev, torch, named=False):
    """fingerprints of the frames [(frame, n)] bucketed by owner -> (per owner: list of uint8 tensors, counts per
    owner[, per owner: the name records of the same pairs in the same order])"""
    n_local = sum(n for _, n in frames)
    send = torch.empty(max(1, n_local) * FP_BYTES, dtype=torch.uint8, device=dev)
    names = torch.empty(max(1, n_local) * NAME_BYTES, dtype=torch.uint8, device=dev) if named else None
    per_frame, off = [], 0
    for fr, n in frames:
        counts = ctx.names_fingerprints(fr, state, base, world, send.data_ptr() + off * FP_BYTES,
                                        names.data_ptr() + off * NAME_BYTES if named else None)
        per_frame.append((off, counts))
        off += sum(counts)
        base += n
    parts, nparts = [[] for _ in range(world)], [[] for _ in range(world)]
    for o0, counts in per_frame:
        p = o0
        for o, c in enumerate(counts):
            parts[o].append(send[p * FP_BYTES:(p + c) * FP_BYTES])
            if named:
                nparts[o].append(names[p * NAME_BYTES:(p + c) * NAME_BYTES])
            p += c
    totals = [sum(c[o] for _, c in per_frame) for o in range(world)]
    return (parts, totals, nparts) if named else (parts, totals)


def paired_by_position(ctx, frames1, state1, base1, frames2, state2, base2, group=None, device=None):
    """Mate files hold their reads in one order.  When every rank holds the SAME records of both files (base1 == base2,
    as many records of each - the shards of a pair cut at the same record numbers) the question "is the name of record i
    of file 2 the name of record i of file 1, for every i" needs no exchange at all: each rank compares its own ranges
    (fqg_frame_name_records / fqg_frame_names_equal, include/fqg.h) and one all-reduce says whether every rank said yes.
    True on every rank, or False on every rank (shards that differ, one name out of place, a name beyond the 56 bytes of
    a name record: the caller exchanges by hash).  With file 1 free of repeated names, True is "every read paired"."""
    import torch
    import torch.distributed as dist

    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    n1 = sum(n for _, n in frames1)
    n2 = sum(n for _, n in frames2)
    ok = base1 == base2 and n1 == n2
    if ok and n1:
        longest = max(n for _, n in list(frames1) + list(frames2))
        buf = torch.empty(longest * NAME_BYTES, dtype=torch.uint8, device=dev)
        a0 = 0
        spans1 = []
        for fr, n in frames1:
            spans1.append((a0, n, fr))
            a0 += n
        b0 = 0
        for fr2, m in frames2:
            for a, n, fr1 in spans1:
                lo, hi = max(a, b0), min(a + n, b0 + m)
                if lo >= hi or not ok:
                    continue
                ctx.frame_name_records(fr1, state1, lo - a, hi - lo, buf.data_ptr())
                eq, _ = ctx.frame_names_equal(fr2, state2, lo - b0, hi - lo, buf.data_ptr())
                ok = ok and eq == hi - lo
            b0 += m
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item())


def global_pairing(ctx, frames1, state1, base1, frames2, state2, base2, group=None, device=None, file1_unique=False):
    """The file-2 loop of fastq_info over ranks that hold ARBITRARY shards of the two files:
    frames1 / frames2 = this rank's retained frames [(frame, n_records)] of file 1 / file 2, whose records are
    base1 / base2, +1, ... in their file.  Every name travels as a 16-byte (fingerprint, index) pair to the
    owner of its fingerprint (one all-to-all over RCCL for both files together); owners sort and classify runs
    of equal fingerprints on the device; what a fingerprint cannot decide is resolved on the name bytes.
    Returns, identically on every rank: dict(matched, leftover, unpaired, first_unpaired = (file-2 record
    index, its name) or None) - first_unpaired is where the serial loop prints "unpaired read", leftover what
    it reports as "found N unpaired reads" when it gets to the end.
    file1_unique: the caller has established that no name of file 1 occurs twice (fastq_info has, by the time it reads
    file 2) - then mates that lie at the same place on every rank are paired without any exchange (paired_by_position;
    FQGPU_NO_POSITIONAL_MATCH=1 leaves that out)."""
    import os

    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    if file1_unique and not os.environ.get("FQGPU_NO_POSITIONAL_MATCH"):
        if paired_by_position(ctx, frames1, state1, base1, frames2, state2, base2, group, dev):
            total = torch.tensor([sum(n for _, n in frames2)], dtype=torch.int64, device=dev if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(total, group=group)
            return {"matched": int(total.item()), "leftover": 0, "unpaired": 0, "first_unpaired": None, "by_position": True}
    p1, c1, n1 = _export_fingerprints(ctx, frames1, state1, base1, world, dev, torch, named=True)
    p2, c2, n2 = _export_fingerprints(ctx, frames2, state2, base2 | FP_FILE2, world, dev, torch, named=True)
    pieces = [t for o in range(world) for t in (p1[o] + p2[o])]
    send = torch.cat(pieces) if pieces else torch.empty(0, dtype=torch.uint8, device=dev)
    npieces = [t for o in range(world) for t in (n1[o] + n2[o])]
    send_names = torch.cat(npieces) if npieces else torch.empty(0, dtype=torch.uint8, device=dev)
    send_counts = [a + b for a, b in zip(c1, c2)]
    # (memory: 16 bytes a pair + 64 a name, sent and received - 160 bytes a record at the peak if everything lived at once;
    # every buffer goes as soon as its exchange is over: 80 + 64 at the peak.  The owner's set copies the names once more)
    recv, recv_counts = exchange_fingerprints(send, send_counts, group)
    del send, pieces, p1, p2
    recv_names, name_counts = exchange_fingerprints(send_names, send_counts, group, record_bytes=NAME_BYTES)
    assert name_counts == recv_counts
    del send_names, npieces, n1, n2
    n_recv = sum(recv_counts)
    fps = ctx.fingerprint_set(max(1024, n_recv))
    try:
        fps.insert(recv.data_ptr(), n_recv, recv_names.data_ptr())
        summary, entries = fps.pair_runs()
    finally:
        fps.close()
    if summary["n_complex"] > len(entries):
        raise RuntimeError(f"{summary['n_complex']} entries need their names: the inputs repeat names massively")
    gathered = [None] * world
    dist.all_gather_object(gathered, (summary, entries), group=group)
    flat = [e for _, ents in gathered for e in ents]
    # every rank names the entries it holds; then everybody resolves identically
    spans = []
    for frames, state, base, flag in ((frames1, state1, base1, 0), (frames2, state2, base2, FP_FILE2)):
        b = base
        for fr, n in frames:
            spans.append((b, n, fr, state, flag))
            b += n

    def local_name(g):
        raw, flag = g & ~FP_FILE2, g & FP_FILE2
        for b0, n, fr, state, fl in spans:
            if fl == flag and b0 <= raw < b0 + n:
                return ctx.frame_name(fr, state, raw - b0)
        return None

    mine = {}
    for _, g in flat:
        if g not in mine:
            nm = local_name(g)
            if nm is not None:
                mine[g] = nm
    names = [None] * world
    dist.all_gather_object(names, mine, group=group)
    table = {}
    for d in names:
        table.update(d)
    parts = [(s["matched"], s["leftover"], s["unpaired"], s["first_unpaired"]) for s, _ in gathered]
    # runs are owner-local: resolve owner by owner (run ids are only unique within one owner)
    for _, ents in gathered:
        if ents:
            parts.append(resolve_pair_runs(ents, table.__getitem__))
    matched, leftover, unpaired, first = merge_pairing(parts)
    out = {"matched": matched, "leftover": leftover, "unpaired": unpaired, "first_unpaired": None}
    if first is not None:
        # its name: held by the rank that holds the record
        nm = local_name(first | FP_FILE2)
        got = [None] * world
        dist.all_gather_object(got, nm, group=group)
        out["first_unpaired"] = (first, next(x for x in got if x is not None))
    return out


# ---- bam_umi_count over shards ---------------------------------------------------------------------
def merge_umi_shards(infos):
    """infos: per rank, in rank order, the dict umi_count(defer_output=True) returned (code, record,
    features, cells, n_alignments, ...).  A CR-sorted file cut at cell boundaries: rank r holds the cells
    that follow those of rank r - 1.  Returns {"finding": (rank, code, record, aux) or None, "remap": per
    rank the local-id -> global-id table of features (index 0 unused), "cell_offset": per rank,
    "features": global names in first-appearance order, "cells": global packed barcodes}.
    Dense ids in order of first appearance (label_str2id / blabel2id, src/bam_umi_count.c:143-260) over
    the whole file = over the ranks in order."""
    for r, info in enumerate(infos):
        if info["code"] != 0:
            return {"finding": (r, info["code"], info["record"], info["aux"])}
    gid, features, remaps = {}, [], []
    for info in infos:
        remap = [0]
        for name in info["features"]:
            if name not in gid:
                features.append(name)
                gid[name] = len(features)
            remap.append(gid[name])
        remaps.append(remap)
    seen, cells, offsets = set(), [], []
    for r, info in enumerate(infos):
        offsets.append(len(cells))
        for c in info["cells"]:
            if c in seen:  # a cell of an earlier shard again: the file is not sorted by cell (:1004-1007)
                return {"finding": (r, 17, None, None)}
            seen.add(c)
            cells.append(c)
    return {"finding": None, "remap": remaps, "cell_offset": offsets, "features": features, "cells": cells}


def unit_float(count):
    """float32 value of `count` additions of 1.0f (saturates at 2**24)"""
    return float(min(count, 1 << 24))


def bam_split(stream):
    """(header bytes, offsets of the alignment records as a numpy array, end of the last complete record) of an
    inflated BAM stream"""
    import ctypes as C

    import numpy as np

    from . import abi as A

    L = A.load()
    buf = (C.c_char * len(stream)).from_buffer_copy(stream)
    n, used = C.c_uint64(), C.c_uint64()
    if L.fqg_bam_index_records(buf, len(stream), None, 0, C.byref(n), C.byref(used)) != 0:
        raise ValueError("not a BAM stream")
    offs = np.zeros(max(1, n.value), dtype=np.uint64)
    L.fqg_bam_index_records(buf, len(stream), offs.ctypes.data_as(C.POINTER(C.c_uint64)), n.value, C.byref(n), C.byref(used))
    offs = offs[:n.value]
    return stream[:int(offs[0]) if n.value else used.value], offs, used.value


def umi_replayed_names(ctx, info):
    """names of the features one of whose (cell, feature) sets the last umi_count(defer_output=True) replayed as the
    reference's RL_Tree behaves: their trees carry state from cell to cell (src/range_list.c:187-198)"""
    flags = ctx.umi_replayed_features(len(info["features"]))
    return [info["features"][f - 1] for f in range(1, len(info["features"]) + 1) if flags[f]]


def umi_records_of(ctx, stream, info, names):
    """the alignment records of the shard just counted (umi_count(defer_output=True) on `stream`) whose feature is one of
    `names`, in file order, as bytes: what a later shard puts in front of its own records so that the trees of these
    features arrive in the state the serial loop would have left them in (src/bam_umi_count.c:418-441: quick_reset_db
    keeps the tree arrays)"""
    import numpy as np

    want = {n for n in names}
    ids = [f for f in range(1, len(info["features"]) + 1) if info["features"][f - 1] in want]
    if not ids:
        return b""
    _, offs, used = bam_split(stream)
    feat = ctx.umi_record_features(len(offs))
    pick = np.nonzero(np.isin(feat, np.array(ids, dtype=np.uint32)))[0]
    ends = np.append(offs[1:], np.uint64(used))
    return b"".join(stream[int(offs[i]):int(ends[i])] for i in pick)


def umi_global_table(umi_lists):
    """umi_lists: per rank, in rank order, the packed UMIs its shard saw in order of first appearance (ctx.umi_umis()).
    The reference numbers UMIs 1, 2, .. in order of first appearance in the WHOLE file (blabel2id, src/bam_umi_count.c:
    225-260) and keeps those NUMBERS in its RL_Tree, so where the tree is not a set the result depends on them: every
    shard must use the file's numbering.  Returns (sorted packed UMIs, their ids) for umi_count(umi_table=...)."""
    import numpy as np

    allv = np.concatenate([np.asarray(x, dtype=np.uint64) for x in umi_lists]) if umi_lists else np.zeros(0, np.uint64)
    if allv.size == 0:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint32)
    keys, first = np.unique(allv, return_index=True)      # first position of every value in rank-then-local order
    ids = np.empty(len(keys), dtype=np.uint32)
    ids[np.argsort(first, kind="stable")] = np.arange(1, len(keys) + 1, dtype=np.uint32)
    return np.ascontiguousarray(keys), np.ascontiguousarray(ids)


def umi_count_behind(ctx, stream, history, **kw):
    """Count a shard with `history` (alignment records of earlier shards, umi_records_of) in front of its own records.
    Returns the counted info, the number of cells the history brought (they come first), and the counters of the
    history alone (to be taken off the totals)."""
    hdr, offs, used = bam_split(stream)
    if not history:
        return ctx.umi_count(stream, defer_output=True, **kw), 0, {"n_new": 0, "n_counted": 0}
    alone = ctx.umi_count(hdr + history, defer_output=True, **kw)
    both = ctx.umi_count(hdr + history + stream[len(hdr):used], defer_output=True, **kw)
    return both, len(alone["cells"]) if alone["code"] == 0 else 0, alone


def umi_finish_shard(ctx, counted, n_history_cells, history_counters, global_ids, cell_offset):
    """output rules for a shard counted by umi_count_behind: feature ids by name from `global_ids`, the cells of the
    history dropped, cell ids continuing at cell_offset + 1"""
    remap = [0] + [global_ids[name] for name in counted["features"]]
    e = ctx.umi_emit(remap, cell_offset - n_history_cells)
    entries = [[t for t in e["entries"][w] if t[1] > cell_offset] for w in range(2)]
    return {"entries": entries, "n_entries": [len(x) for x in entries], "total": [sum(t[2] for t in x) for x in entries],
            "n_new": counted["n_new"] - history_counters["n_new"], "n_counted": counted["n_counted"] - history_counters["n_counted"],
            "rl_undefined": counted.get("rl_undefined", 0) - history_counters.get("rl_undefined", 0)}


def umi_count_sharded(ctx, stream, group=None, **kw):
    """Every rank calls this with ITS shard (an inflated BAM stream holding whole cells, in file order
    over the ranks).
      round 1: every rank counts its shard; the ranks agree on the file's numbering of features, cells and UMIs (one
               all_gather_object of the name / barcode lists: first appearance over the ranks in order);
      round 2: every rank counts again with the file's UMI numbers (the reference's RL_Tree holds those numbers) and
               reports the features one of whose sets is replayed as the tree behaves - a tree carries state from
               cell to cell;
      round 3: every rank puts the alignments of those features from all EARLIER shards (kilobytes per feature) in
               front of its own and counts a last time: each rank then replays exactly what the serial loop replays,
               without waiting for another rank.
    Rounds 2 and 3 only run when there is more than one rank.  A file with fractional increments (NH > 1, several
    genes per alignment) has one more dependence: db->tot_reads_obs / tot_umi_obs are ONE float32 chain over the file
    (src/bam_umi_count.c:490-507), so the last count runs rank after rank, each starting from the totals of the one
    before.  Returns this rank's lines plus the merged header fields; concatenating the lines of the ranks in order
    gives the file."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    local = ctx.umi_count(stream, defer_output=True, **kw)
    infos = [None] * world
    dist.all_gather_object(infos, dict({k: local.get(k) for k in ("code", "record", "aux", "features", "cells", "n_alignments",
                                                                "n_tags_found", "n_umis_discarded", "n_cells_discarded",
                                                                "n_counted", "n_new", "unit_increments", "rl_replayed",
                                                                "tot_reads", "tot_umi")},
                                       umis=ctx.umi_umis() if local["code"] == 0 else None), group=group)
    m = merge_umi_shards(infos)
    if m["finding"] is not None:
        return {"finding": m["finding"]}
    unit = all(i["unit_increments"] for i in infos)
    global_ids = {name: k + 1 for k, name in enumerate(m["features"])}
    counted, n_hist, hist = local, 0, {"n_new": 0, "n_counted": 0}
    chain = None
    if world > 1:
        kw2 = dict(kw)
        history = b""
        if not kw.get("strict_set"):
            kw2["umi_table"] = umi_global_table([i["umis"] for i in infos])
            counted = ctx.umi_count(stream, defer_output=True, **kw2)
            replayed = [None] * world
            dist.all_gather_object(replayed, umi_replayed_names(ctx, counted) if counted["code"] == 0 else [], group=group)
            carried = sorted({n for r in replayed for n in r})
            if carried:
                blobs = [None] * world
                dist.all_gather_object(blobs, umi_records_of(ctx, stream, counted, carried), group=group)
                history = b"".join(blobs[:rank])
        hdr, _, used = bam_split(stream)
        augmented = hdr + history + stream[len(hdr):used] if history else stream
        if history:
            hist = ctx.umi_count(hdr + history, defer_output=True, **kw2)
            n_hist = len(hist["cells"]) if hist["code"] == 0 else 0
        if unit:
            if history:
                counted = ctx.umi_count(augmented, defer_output=True, **kw2)
        else:
            chain = (0.0, 0.0)
            for r in range(world):  # rank after rank: the chain of totals
                box = [None]
                if rank == r:
                    counted = ctx.umi_count(augmented, defer_output=True, db_start=chain,
                                            db_skip=hist.get("n_alignments", 0) if history else 0, **kw2)
                    box = [(counted["tot_reads"], counted["tot_umi"])]
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, r) if group is not None else r, group=group)
                chain = box[0]
    elif not unit:
        chain = (local["tot_reads"], local["tot_umi"])
    mine = umi_finish_shard(ctx, counted, n_hist, hist, global_ids, m["cell_offset"][rank])
    sums = [None] * world
    dist.all_gather_object(sums, (mine["n_entries"], mine["total"], mine["n_new"], mine["n_counted"], mine["rl_undefined"]), group=group)
    return {"finding": None, "entries": mine["entries"], "features": m["features"], "cells": m["cells"],
            "n_entries": [sum(s[0][w] for s in sums) for w in range(2)],
            "total": [sum(s[1][w] for s in sums) for w in range(2)],
            "tot_reads": unit_float(sum(s[3] for s in sums)) if unit else chain[0],
            "tot_umi": unit_float(sum(s[2] for s in sums)) if unit else chain[1],
            "rl_undefined": sum(s[4] for s in sums),
            "n_alignments": sum(i["n_alignments"] for i in infos),
            "n_tags_found": sum(i["n_tags_found"] for i in infos)}
