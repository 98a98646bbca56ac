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
