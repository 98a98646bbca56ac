"""Seeded BAM writer for the bam_umi_count parity tests (test infrastructure; host side only).

Writes what `samtools view -b` would: a BGZF stream (RFC 1952 members with the BC extra field,
SAM/BAM specification section 4) holding the BAM header and alignment records.  Only the fields
bam_umi_count looks at carry meaning (refID, FLAG, the aux tags); the rest is filler."""
import struct
import zlib

import numpy as np


def bgzf(raw: bytes, block=0xFF00, level=6) -> bytes:
    out = []
    for i in range(0, len(raw), block):
        chunk = raw[i:i + block]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = c.compress(chunk) + c.flush()
        bsize = len(comp) + 25
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
                   struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))  # EOF marker
    return b"".join(out)


def header(refs=((b"chr1", 1000000),)) -> bytes:
    text = b"@HD\tVN:1.0\tSO:unsorted\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    h = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs))
    for name, ln in refs:
        h += struct.pack("<i", len(name) + 1) + name + b"\0" + struct.pack("<i", ln)
    return h


def aux_z(tag: bytes, val: bytes) -> bytes:
    return tag + b"Z" + val + b"\0"


def aux_int(tag: bytes, v: int, t=b"C") -> bytes:
    fmt = {b"c": "<b", b"C": "<B", b"s": "<h", b"S": "<H", b"i": "<i", b"I": "<I"}[t]
    return tag + t + struct.pack(fmt, v)


def record(name: bytes, aux: bytes, tid=0, flag=0, pos=100, seq_len=20) -> bytes:
    qn = name + b"\0"
    cigar = struct.pack("<I", seq_len << 4)  # <len>M
    seq = bytes([0x12] * ((seq_len + 1) // 2))
    qual = bytes([30] * seq_len)
    core = struct.pack("<iiIIiiii", tid, pos, (4680 << 16) | (255 << 8) | len(qn), (flag << 16) | 1, seq_len, -1, -1, 0)
    body = core + qn + cigar + seq + qual + aux
    return struct.pack("<i", len(body)) + body


def barcode(rng, n) -> bytes:
    return bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)])


def cells_in_runs(rng, order, per_cell=(3, 40), genes=30):
    """a BAM whose alignments come in runs of one cell each, in the given order of cell numbers"""
    cells = {c: barcode(np.random.default_rng(1000 + c), 12) for c in set(order)}
    out = [header()]
    k = 0
    for c in order:
        for _ in range(int(rng.integers(*per_cell))):
            aux = (aux_z(b"GX", b"GENE%05d" % int(rng.integers(0, genes))) + aux_z(b"RX", barcode(rng, 8)) +
                   aux_z(b"CR", cells[c]) + aux_int(b"NH", 1))
            out.append(record(b"r%d" % k, aux))
            k += 1
    return b"".join(out)


def tagged_bam(rng, n_cells=12, genes=40, reads_per_cell=(5, 120), umi_len=8, cell_len=12, nh=False, multi_gx=False,
               noise=False, sort_cells=True, gene_prefix=b"GENE", fresh_umis=False):
    """A CR-grouped BAM with GX / RX / CR (/ NH) tags.  Returns (bam bytes, inflated record stream).

    fresh_umis: every read carries a UMI never used before in the file, or repeats (gene, UMI) of an
    earlier read of the same cell.  UMI ids then only grow, which keeps the reference's RL_Tree on
    the paths where it behaves as a set ("the reference's UMI set")."""
    recs = []
    cells = []
    while len(cells) < n_cells:
        c = barcode(rng, cell_len)
        if c not in cells:
            cells.append(c)
    order = list(range(n_cells))
    used = set()
    k = 0
    for ci in order:
        n = int(rng.integers(reads_per_cell[0], reads_per_cell[1] + 1))
        umis = [barcode(rng, umi_len) for _ in range(max(1, n // 2))]
        seen_pairs = []
        for _ in range(n):
            g = int(rng.integers(0, genes)) if rng.random() < 0.8 else int(rng.integers(0, max(1, genes // 8)))
            fresh = None
            if fresh_umis:
                if seen_pairs and rng.random() < 0.3:
                    g, fresh = seen_pairs[int(rng.integers(0, len(seen_pairs)))]
                else:
                    if len(used) >= 4 ** umi_len:
                        raise ValueError("fresh_umis: every UMI of %d bases has been used" % umi_len)
                    while fresh is None or fresh in used:
                        fresh = barcode(rng, umi_len)
                    used.add(fresh)
                    seen_pairs.append((g, fresh))
            gx = b"%s%d" % (gene_prefix, g)
            if multi_gx and rng.random() < 0.3:
                r = rng.random()
                if r < 0.3:
                    gx = gx + b"," + gx
                elif r < 0.6:
                    gx = gx + b",%s%d" % (gene_prefix, int(rng.integers(0, genes)))
                elif r < 0.8:
                    gx = b"," + gx + b",,%s%d,%s%d" % (gene_prefix, g + 1, gene_prefix, g + 1)
                else:
                    gx = gx + b"," + gx + b"," + gx
            aux = b""
            flag, tid = 0, 0
            if nh and rng.random() < 0.5:
                aux += aux_int(b"NH", int(rng.integers(1, 5)), [b"C", b"c", b"S", b"i"][int(rng.integers(0, 4))])
            umi = fresh if fresh is not None else umis[int(rng.integers(0, len(umis)))]
            cell = cells[ci]
            if noise:
                r = rng.random()
                if r < 0.04:
                    gx = b""
                elif r < 0.08:
                    umi = b""
                elif r < 0.11:
                    flag = 4
                elif r < 0.14:
                    tid = -1
                elif r < 0.17 and not fresh_umis:
                    umi = umi[:-2].lower() + b"NN"
                elif r < 0.19:
                    aux += aux_int(b"XX", 7, b"i") + b"XBC\x03\x00\x00\x00\x01\x02\x03" + b"XfF\x00\x00\x80\x3f"
            aux += aux_z(b"CR", cell)
            if gx:
                aux += aux_z(b"GX", gx) + aux_z(b"TX", b"T" + gx)
            if umi:
                aux += aux_z(b"RX", umi) + aux_z(b"UB", umi[::-1])
            recs.append(record(b"r%d" % k, aux, tid=tid, flag=flag))
            k += 1
    if not sort_cells:
        perm = rng.permutation(len(recs))
        recs = [recs[i] for i in perm]
    stream = header() + b"".join(recs)
    return bgzf(stream), stream


# ---- large, vectorised: fixed-geometry records (BASELINE.json configs[3]) --------------------------
REC_BYTES = 124


def _digits(a, width):
    out = np.empty((a.size, width), dtype=np.uint8)
    for k in range(width):
        out[:, width - 1 - k] = (a % 10) + 48
        a = a // 10
    return out


def _bases(codes, width):
    """codes: uint64 array, 2 bits per base, most significant base first"""
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = np.empty((codes.size, width), dtype=np.uint8)
    for k in range(width):
        out[:, k] = lut[(codes >> np.uint64(2 * (width - 1 - k))) & np.uint64(3)]
    return out


def fixed_records(cell_code, gene, umi_code):
    """One 124-byte alignment record per element: CR:Z:<16 bases> GX:Z:G<5 digits> RX:Z:<10 bases>.
    Returns a uint8 array of shape (n, 124)."""
    n = cell_code.size
    rec = np.zeros((n, REC_BYTES), dtype=np.uint8)
    core = struct.pack("<iiiIIiiii", REC_BYTES - 4, 0, 100, (4680 << 16) | (255 << 8) | 10, (0 << 16) | 1, 20, -1, -1, 0)
    rec[:, :36] = np.frombuffer(core, dtype=np.uint8)
    rec[:, 36] = ord("r")
    rec[:, 37:45] = _digits(np.arange(n, dtype=np.int64) % 100000000, 8)
    rec[:, 46:50] = np.frombuffer(struct.pack("<I", 20 << 4), dtype=np.uint8)
    rec[:, 50:60] = 0x12
    rec[:, 60:80] = 30
    rec[:, 80:83] = np.frombuffer(b"CRZ", dtype=np.uint8)
    rec[:, 83:99] = _bases(cell_code, 16)
    rec[:, 100:103] = np.frombuffer(b"GXZ", dtype=np.uint8)
    rec[:, 103] = ord("G")
    rec[:, 104:109] = _digits(gene.astype(np.int64), 5)
    rec[:, 110:113] = np.frombuffer(b"RXZ", dtype=np.uint8)
    rec[:, 113:123] = _bases(umi_code, 10)
    return rec


TAGGED_NAME_REC_BYTES = 137


def tagged_name_records(cell_code, umi_code, tid):
    """One 137-byte alignment per element whose read name carries the barcodes as fastq_pre_barcodes writes them:
    STAGS_CELL=<16 bases>_UMI=<10 bases>_SAMPLE=_ETAGS_r<8 digits> (what bam_add_tags parses).  uint8 [n, 137]."""
    n = cell_code.size
    R = TAGGED_NAME_REC_BYTES
    rec = np.zeros((n, R), dtype=np.uint8)
    core = struct.pack("<iiiIIiiii", R - 4, 0, 100, (4680 << 16) | (255 << 8) | 67, (0 << 16) | 1, 20, -1, -1, 0)
    rec[:, :36] = np.frombuffer(core, dtype=np.uint8)
    rec[:, 4:8] = np.asarray(tid, dtype="<i4").reshape(-1, 1).view(np.uint8).reshape(n, 4)
    q = 36
    for text, width, fill in ((b"STAGS_CELL=", 11, None), (None, 16, _bases(cell_code, 16)), (b"_UMI=", 5, None),
                              (None, 10, _bases(umi_code, 10)), (b"_SAMPLE=_ETAGS_r", 16, None),
                              (None, 8, _digits(np.arange(n, dtype=np.int64) % 100000000, 8))):
        rec[:, q:q + width] = np.frombuffer(text, dtype=np.uint8) if text is not None else fill
        q += width
    q += 1  # the name's NUL
    rec[:, q:q + 4] = np.frombuffer(struct.pack("<I", 20 << 4), dtype=np.uint8)
    rec[:, q + 4:q + 14] = 0x12
    rec[:, q + 14:q + 34] = 30
    assert q + 34 == R
    return rec


def config4(rng, n_cells=10000, n_genes=20000, n_triples=5000000, dup=0.3, fresh_umis=False):
    """CR-sorted synthetic alignments: `n_triples` distinct (cell, gene, UMI) plus `dup` duplicate reads.
    fresh_umis: every distinct triple gets a UMI string of its own, in increasing order of first use
    (needs n_triples <= 4**10): the regime in which the reference's RL_Tree behaves as a set.
    Returns (records uint8 [n, 124], cell index, gene, umi code) in file order."""
    cells_code = rng.choice(np.uint64(1) << np.uint64(32), size=n_cells, replace=False).astype(np.uint64)
    cell = np.sort(rng.integers(0, n_cells, n_triples))
    gene = rng.zipf(1.3, n_triples) % n_genes
    if fresh_umis:
        assert n_triples <= 4 ** 10
        umi = rng.permutation(4 ** 10)[:n_triples].astype(np.uint64)
    else:
        umi = rng.integers(0, 4 ** 10, n_triples).astype(np.uint64)
        key = (cell.astype(np.uint64) << np.uint64(40)) | (gene.astype(np.uint64) << np.uint64(20)) | umi
        _, first = np.unique(key, return_index=True)
        first.sort()
        cell, gene, umi = cell[first], gene[first], umi[first]
    n_dup = int(cell.size * dup)
    src = rng.integers(0, cell.size, n_dup)
    cell_all = np.concatenate([cell, cell[src]])
    gene_all = np.concatenate([gene, gene[src]])
    umi_all = np.concatenate([umi, umi[src]])
    # file order: by cell, and inside a cell the originals first in their order (so that with fresh_umis
    # the first use of every UMI happens in increasing order), then the duplicates
    order = np.lexsort((np.arange(cell_all.size), cell_all))
    cell_all, gene_all, umi_all = cell_all[order], gene_all[order], umi_all[order]
    rec = fixed_records(cells_code[cell_all], gene_all, umi_all)
    return rec, cell_all, gene_all, umi_all


def expected_matrix(cell, gene, umi):
    """(cell id, gene id, distinct UMIs, reads) per printed (cell, gene), sorted by cell then gene - the
    answer the counting must give in sorted mode when every increment is 1.0 (numpy, independent of the
    oracle).  Includes the early break of cell2MM (src/bam_umi_count.c:697): a cell prints only the
    features whose id - 1 is smaller than the cell's UMI total."""
    def first_ids(x):
        u, first, inv = np.unique(x, return_index=True, return_inverse=True)
        rank = np.empty(u.size, dtype=np.int64)
        rank[np.argsort(first, kind="stable")] = np.arange(1, u.size + 1)
        return rank[inv], u.size
    cid, n_cells = first_ids(cell)
    gid, n_genes = first_ids(gene)
    pair = cid.astype(np.int64) * (n_genes + 1) + gid
    trip = pair * (4 ** 10) + umi.astype(np.int64)
    up, reads = np.unique(pair, return_counts=True)
    ut = np.unique(trip)
    _, umis = np.unique(ut // (4 ** 10), return_counts=True)
    c, g = up // (n_genes + 1), up % (n_genes + 1)
    tot = np.bincount(c, weights=umis, minlength=n_cells + 1)
    keep = (g - 1) < tot[c]
    return c[keep], g[keep], umis[keep], reads[keep], n_cells, n_genes


def expected_matrix_reference(cell, gene, umi):
    """The same answer as the REFERENCE gives it (sorted mode, unit increments): "new UMI" decided by its
    RL_Tree as it behaves (oracle/rl_oracle.c through oracle.loader.rl_replay), counters and cell2MM's
    rules (src/bam_umi_count.c:444-509, 666-705, 418-441) in numpy.  Returns (lines_u, lines_r, total_u,
    total_r, n_cells, n_genes, stats) with lines as (gene id, cell id, value) in file order."""
    from oracle import loader

    def first_ids(x):
        u, first, inv = np.unique(x, return_index=True, return_inverse=True)
        rank = np.empty(u.size, dtype=np.int64)
        rank[np.argsort(first, kind="stable")] = np.arange(1, u.size + 1)
        return rank[inv], u.size
    cid, n_cells = first_ids(cell)
    gid, n_genes = first_ids(gene)
    uid, _ = first_ids(umi)
    n = cid.size
    is_new, stats = loader.rl_replay(gid, uid, cid, np.ones(n, dtype=np.float32), n_genes + 1)
    pair = cid.astype(np.int64) * (n_genes + 1) + gid
    up, inv, reads = np.unique(pair, return_inverse=True, return_counts=True)
    umis = np.bincount(inv, weights=is_new, minlength=up.size).astype(np.int64)
    c, g = up // (n_genes + 1), up % (n_genes + 1)
    # a feature whose UMI total of a cell is 0 is not reset (quick_reset_db :428-434): its reads would carry over
    # into the next cell.  Every epoch starts on an emptied tree, so its first record is always new.
    assert (umis >= 1).all()
    tot = np.bincount(c, weights=umis, minlength=n_cells + 1)
    keep = (g - 1) < tot[c]
    lu = list(zip(g[keep].tolist(), c[keep].tolist(), umis[keep].tolist()))
    lr = list(zip(g[keep].tolist(), c[keep].tolist(), reads[keep].tolist()))
    return lu, lr, int(umis[keep].sum()), int(reads[keep].sum()), n_cells, n_genes, stats
