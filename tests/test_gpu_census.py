"""FASTQ -> (cell, UMI) without the BAM round trip (fqg_barcodes_census, include/fqg.h).

The checker is the COMPOSITION of what the reference's three programs do to a kept read, each step from a pinned
restatement: the read names fastq_pre_barcodes writes (for the barcode_test2 files: the bytes the reference binary wrote,
tests/golden/pre_barcodes.json; for synthetic 10x-v2 pairs: oracle/pre_barcodes_oracle.py), get_barcodes of bam_add_tags
on those names (oracle/bam_tags_oracle.py, src/bam_add_tags.c:43-99), and char2uint_64 / the "no UMI tag: not counted"
rule of bam_umi_count (oracle/umi_oracle.py, src/bam_umi_count.c:364-382,960)."""
import gzip
import json
import os
from collections import defaultdict

import numpy as np
import pytest

import fastq_utils_amd as fq
from oracle import bam_tags_oracle as bto
from oracle import pre_barcodes_oracle as pbo
from oracle import umi_oracle as uo
from tests.test_gpu_pre_barcodes import make_10x
from tests.util import GOLD

pytestmark = pytest.mark.gpu
A = fq.abi
GOLDEN = json.load(open(os.path.join(GOLD, "pre_barcodes.json")))
FILE_KEYS = {"--read1": A.READ1, "--read2": A.READ2, "--index1": A.INDEX1, "--index2": A.INDEX2, "--index3": A.INDEX3}
REFS = {"read1": A.READ1, "read2": A.READ2, "index1": A.INDEX1, "index2": A.INDEX2, "index3": A.INDEX3}


def expected_census(fastq_text: bytes):
    """names of the reads fastq_pre_barcodes kept -> what bam_add_tags + bam_umi_count make of them"""
    pairs = []
    # (header lines by their tags, not by counting four lines: with --read1_offset 5 --read1_size -1 the reference's
    # slice_read leaves records of two lines, src/fastq_pre_barcodes.c:168-189; a read without tags has no pair anyway)
    for line in fastq_text.split(b"\n"):
        if not line.startswith(b"@STAGS_"):
            continue
        qname = line[1:].split(b" ")[0].split(b"\t")[0] + b"\0"   # what an aligner carries into the BAM record
        ok, cell, umi, _sample = bto.get_barcodes(qname, 0, len(qname))
        if not ok or not umi:          # no RX tag: src/bam_umi_count.c:960
            continue
        pairs.append((uo.char2uint_64(cell), uo.char2uint_64(umi)))   # (no CR tag: char2uint_64 of nothing is 0)
    per_cell = defaultdict(lambda: [0, set()])
    for c, u in pairs:
        per_cell[c][0] += 1
        per_cell[c][1].add(u)
    cells = sorted((c, v[0], len(v[1])) for c, v in per_cell.items())
    return sorted(pairs), cells


def parse(args):
    """the subset of fastq_pre_barcodes' options the census depends on -> files and barcode specs"""
    files, spec, opt = {}, {"umi": [None, -1, 0], "cell": [None, -1, 0], "sample": [None, -1, 0]}, {"min_qual": 0, "phred": 33}
    it = iter(args)
    for a in it:
        if a in FILE_KEYS:
            files[FILE_KEYS[a]] = next(it)
        elif a[2:].split("_")[0] in spec and a.startswith("--"):
            tag, what = a[2:].split("_")
            v = next(it)
            spec[tag][{"read": 0, "offset": 1, "size": 2}[what]] = REFS[v] if what == "read" else int(v)
        elif a == "--min_qual":
            opt["min_qual"] = int(next(it))
        elif a == "--phred_encoding":
            opt["phred"] = int(next(it))
        elif a in ("--outfile1", "--outfile2", "--read1_offset", "--read1_size", "--read2_offset", "--read2_size"):
            next(it)
        else:
            return None   # (--sam, --interleaved, -X ...: not needed here)
    specs = {k: tuple(v) if v[0] is not None else None for k, v in spec.items()}
    return files, specs, opt


def census_of(ctx, images, specs, opt):
    """images: {file reference: FASTQ bytes}.  transform + census -> (pairs, cells, the arrays as the device holds them)"""
    frames, states = {}, {}
    for x, img in images.items():
        st = A.probe_first_record(img, True)
        r = ctx.validate(img, None, st, final=True, flags=A.VALIDATE_FRAME_ONLY)
        assert r["code"] == 0
        frames[x], states[x] = ctx.retain_frame(), st
    n = min(f.n_records for f in frames.values())
    z = ctx.census()
    try:
        kw = dict(umi=specs["umi"], cell=specs["cell"], sample=specs["sample"], phred=opt["phred"], min_qual=opt["min_qual"], sam=False)
        r = ctx.barcodes_transform(frames, states, n, **kw)
        assert r["code"] == 0
        added = ctx.barcodes_census(z, frames, states, r["n_done"], **kw)
        assert added <= r["n_done"] - r["n_discarded"]
        n_pairs, n_cells = z.finish()
        ce, um = z.pairs(n_pairs)
        cells = z.cells(n_cells)
        return sorted(zip(ce.tolist(), um.tolist())), [tuple(int(v) for v in row) for row in cells], (ce, um)
    finally:
        z.close()
        for f in frames.values():
            f.release()


CASES = [i for i, c in enumerate(GOLDEN) if c["exit"] == 0 and "OUT1" in c["files"] and parse(c["args"]) is not None
         and any("barcode_test" in a for a in c["args"])]


@pytest.mark.parametrize("i", CASES)
def test_reference_output_names_give_the_census(i):
    case = GOLDEN[i]
    files, specs, opt = parse(case["args"])
    images = {x: gzip.decompress(open(os.path.join(GOLD, name), "rb").read()) for x, name in files.items()}
    want_pairs, want_cells = expected_census(case["files"]["OUT1"].encode("latin-1"))
    with fq.Context(0) as ctx:
        got_pairs, got_cells, (ce, um) = census_of(ctx, images, specs, opt)
    assert got_pairs == want_pairs
    assert got_cells == want_cells
    # what fqg_census_finish leaves in HBM is sorted by (cell, UMI)
    assert list(zip(ce.tolist(), um.tolist())) == want_pairs


def test_cases_cover_umi_cell_and_sample_layouts():
    assert len(CASES) >= 3
    kinds = set()
    for i in CASES:
        _, specs, _ = parse(GOLDEN[i]["args"])
        kinds.add(tuple(k for k in ("umi", "cell", "sample") if specs[k]))
    assert ("umi",) in kinds and ("umi", "cell") in kinds and ("umi", "cell", "sample") in kinds, kinds


@pytest.mark.parametrize("layout", ["v2", "umi_only", "cell_only", "other_file"])
def test_10x_synthetic_against_the_three_oracles(layout):
    rng = np.random.default_rng(17)
    n = 30000
    r1, r2 = make_10x(rng, n)
    # few cells, so that cells hold many reads and UMIs repeat inside a cell
    lines = r1.split(b"\n")
    pool = [bytes(rng.choice(list(b"ACGT"), 16).astype(np.uint8)) for _ in range(50)]
    upool = [bytes(rng.choice(list(b"ACGTN"), 10).astype(np.uint8)) for _ in range(300)]
    for k in range(n):
        s = lines[4 * k + 1]
        if len(s) == 26:
            lines[4 * k + 1] = pool[int(rng.integers(0, 50))] + (upool[int(rng.integers(0, 300))] if k % 3 else s[16:])
    r1 = b"\n".join(lines)
    specs = {"umi": (A.INDEX1, 16, 10), "cell": (A.INDEX1, 0, 16), "sample": None}
    args = ["--read1", "r2.fastq", "--index1", "r1.fastq", "--phred_encoding", "33", "--min_qual", "10", "--outfile1", "o.fastq.gz"]
    if layout == "umi_only":
        specs["cell"] = None
    if layout == "cell_only":
        specs["umi"] = None      # no UMI: bam_umi_count counts nothing
    if layout == "other_file":
        specs["cell"] = (A.READ1, 3, 12)   # the cell from the cDNA read, the UMI from the index read
    names = {A.READ1: "read1", A.INDEX1: "index1"}
    for tag in ("umi", "cell"):
        if specs[tag]:
            args += ["--%s_read" % tag, names[specs[tag][0]], "--%s_offset" % tag, str(specs[tag][1]), "--%s_size" % tag, str(specs[tag][2])]
    files = {"r1.fastq": r1, "r2.fastq": r2}
    want = pbo.run_pre_barcodes(args, lambda name: files[name])
    assert want["exit"] == 0
    want_pairs, want_cells = expected_census(want["files"][1])
    with fq.Context(0) as ctx:
        got_pairs, got_cells, _ = census_of(ctx, {A.READ1: r2, A.INDEX1: r1}, specs, {"phred": 33, "min_qual": 10})
    assert got_pairs == want_pairs
    assert got_cells == want_cells
    if layout == "cell_only":
        assert got_pairs == [] and got_cells == []
    else:
        assert len(got_cells) >= (1 if layout == "umi_only" else 40)
        if layout != "other_file":   # (a cell from 12 random bases of the cDNA read is nearly every read's own)
            assert any(reads > umis for _, reads, umis in got_cells)   # a UMI seen twice in a cell counts once


def test_census_must_follow_its_transform():
    r1, r2 = make_10x(np.random.default_rng(2), 500)
    with fq.Context(0) as ctx:
        frames, states = {}, {}
        for x, img in ((A.READ1, r2), (A.INDEX1, r1)):
            st = A.probe_first_record(img, True)
            ctx.validate(img, None, st, final=True, flags=A.VALIDATE_FRAME_ONLY)
            frames[x], states[x] = ctx.retain_frame(), st
        z = ctx.census()
        kw = dict(umi=(A.INDEX1, 16, 10), cell=(A.INDEX1, 0, 16), min_qual=10, sam=False)
        with pytest.raises(Exception):
            ctx.barcodes_census(z, frames, states, 500, **kw)   # no transform yet: no status bytes to go by
        r = ctx.barcodes_transform(frames, states, 500, **kw)
        first = ctx.barcodes_census(z, frames, states, r["n_done"], **kw)
        again = ctx.barcodes_census(z, frames, states, r["n_done"], **kw)  # a second batch of the same reads: appended
        assert first == again > 0
        n_pairs, n_cells = z.finish()
        assert n_pairs == 2 * first
        cells = z.cells(n_cells)
        assert int(cells[:, 1].sum()) == n_pairs and (cells[:, 1] >= 2 * cells[:, 2]).all()  # every read twice, every UMI once
        with pytest.raises(Exception):
            ctx.barcodes_census(z, frames, states, r["n_done"], **kw)   # finished
        z.close()
        for f in frames.values():
            f.release()
