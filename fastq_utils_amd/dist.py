"""Multi-GPU plumbing for the record-parallel paths (one process per GPU).

Validation shards trivially: every rank frames and validates its own record-aligned shard; no
data-path collective is needed.  What has to be combined at the end is tiny: per-rank statistics
blobs (fqg_acc_export) and the first finding in file order.  These helpers are pure Python so that
they can be exercised with the gloo backend on CPU.
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
